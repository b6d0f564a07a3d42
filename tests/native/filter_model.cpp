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

// stats: [0] pairs, [1] filter passes, [2] oracle accepts, [3] violations (accept && !pass, either form),
//        [4] passes of the moment form
extern "C" void fm_check_general(int n_tris, const float *verts /* n_tris*9 */, int n_rays,
                                 const float *org /* offset origins */, const float *dir,
                                 const float *center, float radius, long long *stats)
{
    long long pairs = 0, pass = 0, acc = 0, bad = 0, passm = 0;
#pragma omp parallel for reduction(+ : pairs, pass, acc, bad, passm) schedule(static)
    for (int k = 0; k < n_tris; k++) {
        const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
        rtf::TriRec rec;
        rtf::make_general_rec(X, Y, Z, &rec);
        rtf::MomRec mrec;
        rtf::make_moment_rec(X, Y, Z, center, &mrec);
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i, *d = dir + 3 * i;
            float oc[3] = {o[0] - center[0], o[1] - center[1], o[2] - center[2]};
            float reach = norm3(oc) + 3.0f * radius;
            float sc = rtf::dir_scale(reach, d[0], d[1], d[2]);
            float g = rtf::filter_general(rec.v, o[0], o[1], o[2], d[0] * sc, d[1] * sc, d[2] * sc);
            bool p = g >= -1.0f;
            // the moment form of the same filter (the per-pair kernel's): it must let through what the
            // oracle accepts too
            float mx, my, mz;
            rtf::ray_moment(o[0], o[1], o[2], center, d[0] * sc, d[1] * sc, d[2] * sc, mx, my, mz);
            const bool pm = rtf::filter_moment(mrec.v, mx, my, mz, d[0] * sc, d[1] * sc, d[2] * sc) >= -1.0f;
            passm += pm;
            float t;
            bool a = rto_ray_vs_triangle(o, d, X, Y, Z, &t) != 0 && t > 0.0f; // trace() keeps t > 0 only
            pairs++;
            pass += p;
            acc += a;
            bad += (a && !(p && pm));
        }
    }
    stats[0] = pairs; stats[1] = pass; stats[2] = acc; stats[3] = bad;
    stats[4] = passm;
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

// ---------------------------------------------------------------------------------------------
// Clustered traversal of arbitrary rays, as the bounce passes evaluate it: a pair reaches the
// reference arithmetic iff
//   (its triangle's normal group holds the ray's direction AND the triangle's own band and slab
//    hold the ray, or the ray is outside the records' range)  AND the general edge filter passes,
//   OR  (the ray is inside the range AND its walk over the cells in space (rt_voxel.h) meets a cell
//        that lists the triangle)  AND the general edge filter passes.
// Checked twice: pair by pair with a walk that never stops early (every pair the oracle accepts must
// reach the reference arithmetic), and ray by ray with the walk's stopping rule, the best hit updated
// the moment a candidate is confirmed (the earliest any implementation can stop): the closest hit
// (t bits, lowest index among equal t) must be the one the oracle's sequential loop keeps.
// stats: [0] pairs [1] (ray, group) band passes x members [2] member band+slab passes
//        [3] pairs the never-stopping walk visits [4] pairs reaching the reference arithmetic
//        [5] oracle accepts [6] violations, pair by pair [7] rays outside the records' range
//        [8] triangles that are in no cell [9] triangles that are never accepted
//        [10] rays whose closest hit differs with the stopping rule [11] cells the stopping walks visit
//        [12] triangles they put through the edge filter [13] rays that hit something
// ---------------------------------------------------------------------------------------------
#include "../../opengl_raytracer_amd/csrc/rt_cluster.h"
#include "../../opengl_raytracer_amd/csrc/rt_voxel.h"

extern "C" void fm_check_clustered(int n_tris, const float *verts, int n_rays, const float *org, const float *dir,
                                   const float *center, float radius, int voxel_res, long long *stats)
{
    std::vector<rt_vertex> v((size_t)n_tris * 3);
    for (int i = 0; i < n_tris * 3; i++) {
        v[i] = rt_vertex{};
        v[i].pos[0] = verts[3 * i]; v[i].pos[1] = verts[3 * i + 1]; v[i].pos[2] = verts[3 * i + 2];
    }
    rtc::ClusterSet cs;
    rtc::build(v.data(), n_tris, center, radius, cs);
    rtv::VoxelSet vs;
    const double abs_extent = std::sqrt((double)center[0] * center[0] + (double)center[1] * center[1] +
                                        (double)center[2] * center[2]) + 3.0 * (double)radius;
    rtv::build(v.data(), n_tris, cs.rho, abs_extent, voxel_res, vs);
    const rtv::Grid &G = vs.grid;
    std::vector<int> group_of(n_tris, -1), gpos(n_tris, -1);
    for (int g = 0; g < cs.num_ngroups; g++)
        for (int j = 0; j < rtc::NGROUP; j++) {
            const int t = cs.nmembers[(size_t)g * rtc::NGROUP + j];
            if (t >= 0) { group_of[t] = g; gpos[t] = g * rtc::NGROUP + j; }
        }
    std::vector<rtf::TriRec> grec(n_tris);
    for (int k = 0; k < n_tris; k++) rtf::make_general_rec(verts + 9 * k, verts + 9 * k + 3, verts + 9 * k + 6, &grec[k]);
    long long pairs = 0, gp = 0, mp = 0, sp = 0, reach_exact = 0, acc = 0, bad = 0, wildn = 0, noclu = 0, never = 0;
    long long bad_closest = 0, cells_n = 0, tests_n = 0, hits_n = 0;
    for (int k = 0; k < n_tris; k++) {
        noclu += !(cs.rho[k] >= 0.0);
        never += group_of[k] < 0;
    }
#pragma omp parallel reduction(+ : pairs, gp, mp, sp, reach_exact, acc, bad, wildn, bad_closest, cells_n, tests_n, hits_n)
    {
        std::vector<char> visited(n_tris, 0), band_exact(n_tris, 0);
        std::vector<int> touched;
#pragma omp for schedule(dynamic, 16)
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i, *d = dir + 3 * i;
            // prepare_general
            const float ocx = o[0] - center[0], ocy = o[1] - center[1], ocz = o[2] - center[2];
            const float dist = sqrtf(ocx * ocx + ocy * ocy + ocz * ocz);
            const float g0 = rtf::dir_scale(dist + 3.0f * radius, d[0], d[1], d[2]);
            const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
            const float inv = 1.0f / sqrtf(dd);
            const float bx = d[0] * inv, by = d[1] * inv, bz = d[2] * inv;
            const float g = (dist <= 1.25f * radius && dd > 1e-30f && dd < 1e30f) ? g0 : -g0;
            const bool wild = !(g > 0.0f);
            const bool walks = g > 0.0f;
            const float gg = fabsf(g);
            wildn += wild;
            // the walk that never stops: every triangle of every cell on the line
            touched.clear();
            if (walks) {
                rtv::Walk w{};
                if (rtv::walk_start(G, o[0], o[1], o[2], d[0], d[1], d[2], w)) {
                    do {
                        const int cell = rtv::cell_index(G, w);
                        const int at = cell * rtv::CELL_STRIDE + rtv::LIST_START; // every triangle of the cell
                        for (uint32_t e = vs.off[at]; e < vs.off[at + 1]; e++) {
                            const int t = vs.entries[e];
                            if (!visited[t]) { visited[t] = 1; touched.push_back(t); }
                        }
                    } while (rtv::walk_step(G, o[0], o[1], o[2], w));
                }
            }
            float brute_t = INFINITY;
            int brute_id = -1;
            float model_t = INFINITY;
            int model_id = -1;
            for (int k = 0; k < n_tris; k++) {
                const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
                bool to_exact = false;
                if (group_of[k] >= 0) {
                    const float *Gr = &cs.ngroups[(size_t)group_of[k] * 4];
                    const float tg = fabsf(fmaf(Gr[2], bz, fmaf(Gr[1], by, Gr[0] * bx)));
                    // the part of the group the member sits in (traverse_band_kernel tests it too)
                    const float *Q = &cs.nsubs[(size_t)(gpos[k] / rtc::SUBGROUP) * 4];
                    const float tq = fabsf(fmaf(Q[2], bz, fmaf(Q[1], by, Q[0] * bx)));
                    if ((tg < Gr[3] && tq < Q[3]) || wild) {
                        gp++;
                        const float *m = &cs.nrecs[(size_t)gpos[k] * 8];
                        const float t = fmaf(m[2], bz, fmaf(m[1], by, m[0] * bx));
                        const float hgt = fmaf(m[6], o[2], fmaf(m[5], o[1], fmaf(m[4], o[0], -m[3])));
                        const bool inside = fmaxf(fabsf(t), fabsf(hgt)) < 1.0f;
                        if (wild ? (m[3] != 2.0f) : inside) {
                            mp++;
                            to_exact = rtf::filter_general(grec[k].v, o[0], o[1], o[2], d[0] * gg, d[1] * gg, d[2] * gg) >= -1.0f;
                        }
                    }
                }
                band_exact[k] = to_exact;
                if (!to_exact && visited[k]) {
                    sp++;
                    to_exact = rtf::filter_general(grec[k].v, o[0], o[1], o[2], d[0] * gg, d[1] * gg, d[2] * gg) >= -1.0f;
                }
                float t;
                const bool a = rto_ray_vs_triangle(o, d, X, Y, Z, &t) != 0 && t > 0.0f;
                pairs++;
                reach_exact += to_exact;
                acc += a;
                bad += (a && !to_exact);
                if (a && t < brute_t) { brute_t = t; brute_id = k; } // the reference's sequential rule
                if (a && band_exact[k] && (t < model_t || (t == model_t && k < model_id))) { model_t = t; model_id = k; }
            }
            for (int t : touched) visited[t] = 0;
            // the walk with the stopping rule
            if (walks) {
                rtv::Walk w{};
                float tbest = INFINITY;
                int idbest = -1;
                if (rtv::walk_start(G, o[0], o[1], o[2], d[0], d[1], d[2], w)) {
                    for (;;) {
                        cells_n++;
                        // (as the kernel walks: the list of what the cell holds beyond the cell just left)
                        const int at = rtv::list_index(G, w);
                        for (uint32_t e = vs.off[at]; e < vs.off[at + 1]; e++) {
                            const int k = vs.entries[e];
                            tests_n++;
                            if (!(rtf::filter_general(grec[k].v, o[0], o[1], o[2], d[0] * gg, d[1] * gg, d[2] * gg) >= -1.0f))
                                continue;
                            float t;
                            if (rto_ray_vs_triangle(o, d, verts + 9 * k, verts + 9 * k + 3, verts + 9 * k + 6, &t) != 0 &&
                                t > 0.0f && (t < tbest || (t == tbest && k < idbest))) {
                                tbest = t;
                                idbest = k;
                            }
                        }
                        if (!rtv::walk_step(G, o[0], o[1], o[2], w) || rtv::walk_done(G, w, tbest)) break;
                    }
                }
                if (idbest >= 0 && (tbest < model_t || (tbest == model_t && idbest < model_id))) { model_t = tbest; model_id = idbest; }
            }
            hits_n += brute_id >= 0;
            uint32_t tb, tm;
            memcpy(&tb, &brute_t, 4);
            memcpy(&tm, &model_t, 4);
            bad_closest += (brute_id != model_id) || (tb != tm);
        }
    }
    stats[0] = pairs; stats[1] = gp; stats[2] = mp; stats[3] = sp; stats[4] = reach_exact; stats[5] = acc;
    stats[6] = bad; stats[7] = wildn; stats[8] = noclu; stats[9] = never;
    stats[10] = bad_closest; stats[11] = cells_n; stats[12] = tests_n; stats[13] = hits_n;
}

// Cone of a run of 32 triangles (rt_filter.h make_cluster_cone) and the cone around four such cones
// against the cones of the members: whenever a member's float test passes for a direction, the
// run's test and the test of the level above must pass too.
// stats: [0] (ray, run) pairs [1] run passes [2] (ray, member) passes [3] violations
//        [4] runs whose cone is "everything" [5] runs whose cone is "nothing"
extern "C" void fm_check_cluster_cone(int n_tris, const float *verts, int n_rays, const float *O, const float *dir,
                                      const float *center, float radius, float sign, long long *stats)
{
    const double extent = rtf::scene_extent(O, center, radius);
    const double rmax = rtf::cone_reach_max(O, center, radius);
    const int runs = (n_tris + 31) / 32;
    std::vector<rtf::ConeRec> cones((size_t)runs * 32), rc(runs);
    for (int k = 0; k < runs * 32; k++) {
        if (k >= n_tris) { rtf::cone_never(&cones[k]); continue; }
        const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
        rtf::TriRec rec;
        rtf::make_origin_rec(X, Y, Z, O, &rec);
        rtf::make_cone_rec(X, Y, Z, O, &rec, extent, rmax, &cones[k]);
    }
    long long all = 0, never = 0;
    for (int c = 0; c < runs; c++) {
        rtf::make_cluster_cone(&cones[(size_t)c * 32], 32, &rc[c]);
        all += rc[c].v[3] == -2.0f;
        never += rc[c].v[3] == 2.0f;
    }
    // the level above, as the traversal builds it: a cone around four run cones
    const int supers = (runs + 3) / 4;
    std::vector<rtf::ConeRec> rc4((size_t)supers * 4), sc(supers);
    for (int c = 0; c < supers * 4; c++) {
        if (c < runs) rc4[c] = rc[c];
        else rtf::cone_never(&rc4[c]);
    }
    for (int q = 0; q < supers; q++) rtf::make_cluster_cone(&rc4[(size_t)q * 4], 4, &sc[q]);
    long long pairs = 0, rp = 0, mp = 0, bad = 0;
#pragma omp parallel for reduction(+ : pairs, rp, mp, bad) schedule(static)
    for (int i = 0; i < n_rays; i++) {
        const float *d = dir + 3 * i;
        const float inv = sign / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float bx = d[0] * inv, by = d[1] * inv, bz = d[2] * inv;
        for (int c = 0; c < runs; c++) {
            const float tr = fmaf(rc[c].v[2], bz, fmaf(rc[c].v[1], by, rc[c].v[0] * bx));
            const rtf::ConeRec &su = sc[c / 4];
            const bool super_pass = fmaf(su.v[2], bz, fmaf(su.v[1], by, su.v[0] * bx)) >= su.v[3];
            const bool run_pass = (tr >= rc[c].v[3]) && super_pass;
            pairs++;
            rp += run_pass;
            for (int j = 0; j < 32; j++) {
                const rtf::ConeRec &m = cones[(size_t)c * 32 + j];
                const float t = fmaf(m.v[2], bz, fmaf(m.v[1], by, m.v[0] * bx));
                const bool in = (t >= m.v[3]) && (m.v[3] < 2.0f);
                mp += in;
                bad += (in && !run_pass);
            }
        }
    }
    stats[0] = pairs; stats[1] = rp; stats[2] = mp; stats[3] = bad; stats[4] = all; stats[5] = never;
}

// ---------------------------------------------------------------------------------------------
// Direction cells (rt_grid.h), as grid_build_kernel + traverse_grid_kernel evaluate them: a pair
// reaches the reference arithmetic iff the slot is on the list of the ray's cell -- its cone record
// and its three edge planes meet the cell and every coarser cell above it -- or on the list
// "everywhere", or the ray is outside the cones' validity (wild), AND the edge filter passes.
// stats: [0] pairs [1] pairs on the cell's list [2] pairs reaching the reference arithmetic
//        [3] oracle accepts [4] violations [5] pairs the cone + edge filter cascade lets through
//        [6] wild rays [7] pairs where the record-based tests and the slot view disagree (must be 0)
// ---------------------------------------------------------------------------------------------
#include "../../opengl_raytracer_amd/csrc/rt_grid.h"

extern "C" void fm_check_grid(int n_tris, const float *verts, int n_rays, const float *O, const float *org,
                              const float *dir, const float *tmax, const float *center, float radius, float sign,
                              int bits, long long *stats)
{
    const double extent = rtf::scene_extent(O, center, radius);
    const double rmax = rtf::cone_reach_max(O, center, radius);
    const float rmaxf = (float)rmax;
    const int levels = bits - rtg::MIN_LEVEL + 1;
    // per ray: unit direction, scale, the cones of its cell at every level
    std::vector<rtg::CellCone> cc((size_t)n_rays * levels);
    std::vector<float> bb((size_t)n_rays * 4);
    std::vector<char> wild(n_rays);
    long long wildn = 0;
    for (int i = 0; i < n_rays; i++) {
        const float *o = org + 3 * i, *d = dir + 3 * i;
        const float reach = rtf::reach_shared(O, o, center, radius);
        const float inv = sign / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        const float bx = d[0] * inv, by = d[1] * inv, bz = d[2] * inv;
        bb[4 * i] = bx; bb[4 * i + 1] = by; bb[4 * i + 2] = bz;
        bb[4 * i + 3] = rtf::dir_scale(reach, bx, by, bz);
        wild[i] = !(reach <= rmaxf);
        wildn += wild[i];
        const uint32_t cell = rtg::oct_cell(bx, by, bz, bits);
        for (int l = rtg::MIN_LEVEL; l <= bits; l++)
            rtg::make_cell_cone(cell >> (2 * (bits - l)), l, &cc[(size_t)i * levels + (l - rtg::MIN_LEVEL)]);
    }
    long long pairs = 0, listed_n = 0, p2 = 0, acc = 0, bad = 0, old_n = 0, differ = 0;
#pragma omp parallel for reduction(+ : pairs, listed_n, p2, acc, bad, old_n, differ) schedule(static)
    for (int k = 0; k < n_tris; k++) {
        const float *X = verts + 9 * k, *Y = X + 3, *Z = X + 6;
        rtf::TriRec rec;
        rtf::ConeRec cone;
        rtf::make_origin_rec(X, Y, Z, O, &rec);
        rtf::make_cone_rec(X, Y, Z, O, &rec, extent, rmax, &cone);
        rtg::SlotView view;
        rtg::make_slot_view(cone.v, X, Y, Z, O, &rec, extent, rmax, &view);
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i, *d = dir + 3 * i;
            const float bx = bb[4 * i], by = bb[4 * i + 1], bz = bb[4 * i + 2], g = fabsf(bb[4 * i + 3]);
            const rtg::CellCone *chain = &cc[(size_t)i * levels];
            bool listed = view.cone != 0;
            if (listed && !rtg::slot_everywhere(view))
                for (int l = 0; l < levels && listed; l++) listed = rtg::cell_meets_slot(chain[l], view);
            {
                const rtg::CellCone &leaf = chain[levels - 1];
                const bool by_records = rtg::cell_overlaps(leaf, cone.v[0], cone.v[1], cone.v[2], cone.v[3]) &&
                                        rtg::cell_meets_edges(leaf, X, Y, Z, O, &rec, extent, rmax);
                differ += by_records != rtg::cell_meets_slot(leaf, view);
            }
            const bool edge = rtf::filter_origin(rec.v, bx * g, by * g, bz * g) >= -1.0f;
            const bool reach_exact = (listed || wild[i]) && edge;
            const float dt = fmaf(cone.v[2], bz, fmaf(cone.v[1], by, cone.v[0] * bx));
            const bool old_path = ((dt >= cone.v[3]) || wild[i]) && edge;
            float t;
            bool a = rto_ray_vs_triangle(o, d, X, Y, Z, &t) != 0 && t > 0.0f;
            if (a && tmax) a = t <= tmax[i];
            pairs++;
            listed_n += listed;
            p2 += reach_exact;
            old_n += old_path;
            acc += a;
            bad += (a && !reach_exact);
        }
    }
    stats[0] = pairs; stats[1] = listed_n; stats[2] = p2; stats[3] = acc; stats[4] = bad; stats[5] = old_n;
    stats[6] = wildn; stats[7] = differ;
}

// ---------------------------------------------------------------------------------------------
// Direction cells of the band sweep, as band_build_kernel + traverse_bandcell_kernel evaluate them:
// a member is on the list of the ray's (folded) cell iff its band meets the cell and every coarser
// cell above it, all of which reach the hemisphere z >= 0; members whose band is every direction are
// on the list "everywhere".  A pair the member test  max(|n1.b|, |n2.o - kappa'|) < 1  lets through
// must be listed.  With G > 1 the cell's list is split into G x G origin buckets (rt_grid.h): the member
// must be on the list of the ray's bucket; a ray whose origin has no bucket reads the cell's whole list.
// Every bucket a member is listed in must also lie inside bucket_column_range(), which is what the build
// kernel loops over.
// stats: [0] pairs [1] listed [2] member test passes [3] violations (passes && !listed)
//        [4] pairs of rays without a bucket [5] listed buckets outside their column's range
// ---------------------------------------------------------------------------------------------
extern "C" void fm_check_band_cells(int n_tris, const float *verts, int n_rays, const float *org, const float *dir,
                                    const float *center, float radius, int bits, int G, long long *stats)
{
    rtc::ClusterSet cs;
    std::vector<rt_vertex> v((size_t)n_tris * 3);
    for (int i = 0; i < n_tris * 3; i++) {
        memset(&v[i], 0, sizeof(rt_vertex));
        for (int a = 0; a < 3; a++) v[i].pos[a] = verts[3 * i + a];
    }
    rtc::build(v.data(), n_tris, center, radius, cs);
    const int levels = bits - rtg::MIN_LEVEL + 1;
    std::vector<rtg::CellCone> cc((size_t)n_rays * levels);
    std::vector<float> bb((size_t)n_rays * 3);
    // the mesh's box relative to the centre, with room, as build_band_cells() in rt_api.hip forms it
    double blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0};
    {
        bool first = true;
        for (int i = 0; i < n_tris * 3; i++) {
            const float *q = verts + 3 * i;
            if (!(std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2]))) continue;
            for (int a = 0; a < 3; a++) {
                if (first || q[a] < blo[a]) blo[a] = q[a];
                if (first || q[a] > bhi[a]) bhi[a] = q[a];
            }
            first = false;
        }
        for (int a = 0; a < 3; a++) {
            const double room = 0.004 + 1e-3 * (bhi[a] - blo[a]) + 1e-5 * std::fabs((double)center[a]);
            blo[a] = blo[a] - room - (double)center[a];
            bhi[a] = bhi[a] + room - (double)center[a];
        }
    }
    std::vector<rtg::BucketFrame> frame(n_rays);
    std::vector<int> bucket(n_rays, 0);
    for (int i = 0; i < n_rays; i++) {
        const float *d = dir + 3 * i;
        const float inv = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        bb[3 * i] = d[0] * inv; bb[3 * i + 1] = d[1] * inv; bb[3 * i + 2] = d[2] * inv;
        const uint32_t cell = rtg::oct_cell_folded(bb[3 * i], bb[3 * i + 1], bb[3 * i + 2], bits);
        for (int l = rtg::MIN_LEVEL; l <= bits; l++)
            rtg::make_cell_cone(cell >> (2 * (bits - l)), l, &cc[(size_t)i * levels + (l - rtg::MIN_LEVEL)]);
        if (G > 1) {
            const float *o = org + 3 * i;
            rtg::bucket_frame(cc[(size_t)i * levels + levels - 1], blo, bhi, &frame[i]);
            bucket[i] = rtg::bucket_of(frame[i], o[0] - center[0], o[1] - center[1], o[2] - center[2], G);
        }
    }
    const size_t members = cs.nmembers.size();
    long long pairs = 0, listed_n = 0, pass_n = 0, bad = 0, no_bucket = 0, off_range = 0;
#pragma omp parallel for reduction(+ : pairs, listed_n, pass_n, bad, no_bucket, off_range) schedule(static)
    for (size_t pos = 0; pos < members; pos++) {
        const float *m1 = &cs.nrecs[pos * 8], *m2 = m1 + 4;
        int32_t tri;
        memcpy(&tri, &m2[3], 4);
        if (tri < 0 || m1[3] == 2.0f) continue;
        const bool everywhere = m1[0] == 0.0f && m1[1] == 0.0f && m1[2] == 0.0f;
        for (int i = 0; i < n_rays; i++) {
            const float *o = org + 3 * i;
            const float bx = bb[3 * i], by = bb[3 * i + 1], bz = bb[3 * i + 2];
            no_bucket += bucket[i] < 0;
            bool listed = true;
            if (!everywhere) {
                for (int l = 0; l < levels && listed; l++) {
                    const rtg::CellCone &c = cc[(size_t)i * levels + l];
                    listed = rtg::cell_reaches_upper_hemisphere(c) && rtg::cell_meets_band(c, m1[0], m1[1], m1[2], 1.0f);
                }
                if (listed && G > 1 && bucket[i] >= 0) {
                    const rtg::BucketStrip st = rtg::bucket_strip(frame[i], m2[0], m2[1], m2[2], m1[3], center, G);
                    listed = rtg::bucket_meets_strip(st, bucket[i] % G, bucket[i] / G);
                    if (listed) {
                        int lo, hi;
                        rtg::bucket_column_range(st, bucket[i] % G, G, &lo, &hi);
                        off_range += !(bucket[i] / G >= lo && bucket[i] / G <= hi);
                    }
                }
            }
            const float t0 = fmaf(m1[2], bz, fmaf(m1[1], by, m1[0] * bx));
            const float h0 = fmaf(m2[2], o[2], fmaf(m2[1], o[1], fmaf(m2[0], o[0], -m1[3])));
            const bool pass = fmaxf(fabsf(t0), fabsf(h0)) < 1.0f;
            pairs++;
            listed_n += listed;
            pass_n += pass;
            bad += (pass && !listed);
        }
    }
    stats[0] = pairs; stats[1] = listed_n; stats[2] = pass_n; stats[3] = bad; stats[4] = no_bucket; stats[5] = off_range;
}

// ---------------------------------------------------------------------------------------------
// bucket_column_range() against bucket_meets_strip() over every bucket: the band_bucket_kernel loops over
// the range only, so a bucket that meets the strip outside it would be a member missing from a list.
// strips: n x 8 doubles (k, a, b, bound, u0, v0, su, sv).
// stats: [0] buckets that meet their strip [1] of them outside the column's range [2] buckets in ranges
// ---------------------------------------------------------------------------------------------
extern "C" void fm_check_bucket_columns(int n, const double *strips, int G, long long *stats)
{
    long long meet = 0, bad = 0, ranged = 0;
    for (int i = 0; i < n; i++) {
        const double *q = strips + 8 * i;
        rtg::BucketStrip st;
        st.k = q[0]; st.a = q[1]; st.b = q[2]; st.bound = q[3]; st.u0 = q[4]; st.v0 = q[5]; st.su = q[6]; st.sv = q[7];
        for (int ix = 0; ix < G; ix++) {
            int lo, hi;
            rtg::bucket_column_range(st, ix, G, &lo, &hi);
            if (hi >= lo) ranged += hi - lo + 1;
            if (lo < 0 || hi > G - 1) bad++;
            for (int iy = 0; iy < G; iy++)
                if (rtg::bucket_meets_strip(st, ix, iy)) {
                    meet++;
                    bad += !(iy >= lo && iy <= hi);
                }
        }
    }
    stats[0] = meet; stats[1] = bad; stats[2] = ranged;
}

// ---------------------------------------------------------------------------------------------
// The cells of a walk by rank (rt_voxel.h: event_rank and friends, what traverse_voxel_long_kernel's 64
// lanes compute together) against the cells walk_step() finds one after the other, on a grid given by
// its corner, cell size and cell counts, round after round of 64 steps until the walk leaves the grid:
// every state's cell and entry parameter, the step that leaves the grid, and the state a round hands to
// the next (indices and the three exit parameters), bit for bit.
// stats: [0] walks [1] states compared [2] mismatches [3] rounds [4] walks that never entered the grid
// ---------------------------------------------------------------------------------------------
extern "C" void fm_check_walk_ranks(const float *gmin, float cs, const int *n, float back, float flat, int n_rays,
                                    const float *org, const float *dir, long long *stats)
{
    rtv::Grid G{};
    G.gminx = gmin[0]; G.gminy = gmin[1]; G.gminz = gmin[2];
    G.cs = cs; G.inv_cs = 1.0f / cs;
    G.nx = n[0]; G.ny = n[1]; G.nz = n[2];
    G.gmaxx = rtv::plane_at(G, G.gminx, G.nx); G.gmaxy = rtv::plane_at(G, G.gminy, G.ny); G.gmaxz = rtv::plane_at(G, G.gminz, G.nz);
    G.back = back; G.slack = 0.0f; G.flat = flat;
    long long walks = 0, states = 0, bad = 0, rounds = 0, outside = 0;
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : walks, states, bad, rounds, outside)
    for (int i = 0; i < n_rays; i++) {
        const float *o = org + 3 * i, *d = dir + 3 * i;
        rtv::Walk w{};
        if (!rtv::walk_start(G, o[0], o[1], o[2], d[0], d[1], d[2], w)) { outside++; continue; }
        walks++;
        for (bool over = false; !over;) {
            rounds++;
            // one after the other: states 0..64 from here, and the step that fails
            rtv::Walk seq[65];
            int fail_step = 1 << 20;
            seq[0] = w;
            for (int k = 0; k < 64; k++) {
                seq[k + 1] = seq[k];
                if (!rtv::walk_step(G, o[0], o[1], o[2], seq[k + 1])) { fail_step = k; break; }
            }
            // by rank
            rtv::AxisSeq sx, sy, sz;
            rtv::walk_sequences(G, o[0], o[1], o[2], d[0], d[1], d[2], w, sx, sy, sz);
            int cell[65], nxt[3] = {0, 0, 0}, filled[65] = {0};
            float enter[65];
            cell[0] = rtv::cell_index(G, w); enter[0] = w.t_enter; filled[0] = 1;
            int rank[3][64];
            for (int L = 0; L < 64; L++) {
                float te; int cb, cc;
                int r = rtv::event_rank(G, sx, L, sy, false, sz, false, te, cb, cc);
                rank[0][L] = r;
                if (r < 64) {
                    const int ix = rtv::seq_index(sx, L + 1), iy = rtv::seq_index(sy, cb), iz = rtv::seq_index(sz, cc);
                    cell[r + 1] = (iz * G.ny + iy) * G.nx + ix; enter[r + 1] = te; filled[r + 1]++;
                    if (r == 63) { nxt[0] = ix; nxt[1] = iy; nxt[2] = iz; }
                }
                r = rtv::event_rank(G, sy, L, sx, true, sz, false, te, cb, cc);
                rank[1][L] = r;
                if (r < 64) {
                    const int ix = rtv::seq_index(sx, cb), iy = rtv::seq_index(sy, L + 1), iz = rtv::seq_index(sz, cc);
                    cell[r + 1] = (iz * G.ny + iy) * G.nx + ix; enter[r + 1] = te; filled[r + 1]++;
                    if (r == 63) { nxt[0] = ix; nxt[1] = iy; nxt[2] = iz; }
                }
                r = rtv::event_rank(G, sz, L, sx, true, sy, true, te, cb, cc);
                rank[2][L] = r;
                if (r < 64) {
                    const int ix = rtv::seq_index(sx, cb), iy = rtv::seq_index(sy, cc), iz = rtv::seq_index(sz, L + 1);
                    cell[r + 1] = (iz * G.ny + iy) * G.nx + ix; enter[r + 1] = te; filled[r + 1]++;
                    if (r == 63) { nxt[0] = ix; nxt[1] = iy; nxt[2] = iz; }
                }
            }
            const int ex = rtv::seq_exit_event(sx), ey = rtv::seq_exit_event(sy), ez = rtv::seq_exit_event(sz);
            int r_exit = 1 << 20;
            if (ex < 64) r_exit = std::min(r_exit, rank[0][ex]);
            if (ey < 64) r_exit = std::min(r_exit, rank[1][ey]);
            if (ez < 64) r_exit = std::min(r_exit, rank[2][ez]);
            if (r_exit >= 64) r_exit = 1 << 20;
            bad += r_exit != fail_step;
            const int last = fail_step < 64 ? fail_step : 64; // states 0..last exist
            for (int k = 0; k <= last; k++) {
                states++;
                bad += filled[k] != 1 || cell[k] != rtv::cell_index(G, seq[k]) || bits(enter[k]) != bits(seq[k].t_enter);
            }
            if (fail_step < 64) { over = true; continue; }
            rtv::Walk nw = w;
            nw.ix = nxt[0]; nw.iy = nxt[1]; nw.iz = nxt[2];
            nw.t_enter = enter[64];
            sx.i0 = nw.ix; sy.i0 = nw.iy; sz.i0 = nw.iz;
            nw.tx = rtv::seq_t(G, sx, 0); nw.ty = rtv::seq_t(G, sy, 0); nw.tz = rtv::seq_t(G, sz, 0);
            const rtv::Walk &ref = seq[64];
            bad += nw.ix != ref.ix || nw.iy != ref.iy || nw.iz != ref.iz || bits(nw.tx) != bits(ref.tx) ||
                   bits(nw.ty) != bits(ref.ty) || bits(nw.tz) != bits(ref.tz) || bits(nw.t_enter) != bits(ref.t_enter);
            w = nw;
        }
    }
    stats[0] = walks; stats[1] = states; stats[2] = bad; stats[3] = rounds; stats[4] = outside;
}

// ---------------------------------------------------------------------------------------------
// The seven lists of every cell in space (rt_voxel.h).  Checked against their definition, cell by cell:
// list 6 of a cell is exactly the set of triangles whose box / triangle test against the cell passes
// (tested here once more, independently of the order the build files them in); list f (0..5) is exactly
// list 6 without what list 6 of the face neighbour f holds (everything, when that neighbour is outside the
// grid); a triangle stands on a list once; the offsets are those of the lists laid end to end.
// stats: [0] cells [1] entries of all lists 6 [2] entries of all lists 0..5 [3] violations
// ---------------------------------------------------------------------------------------------
extern "C" void fm_check_cell_lists(int n_tris, const float *verts, const float *center, float radius, int voxel_res,
                                    long long *stats)
{
    std::vector<rt_vertex> v((size_t)n_tris * 3);
    for (int i = 0; i < n_tris * 3; i++) {
        v[i] = rt_vertex{};
        v[i].pos[0] = verts[3 * i]; v[i].pos[1] = verts[3 * i + 1]; v[i].pos[2] = verts[3 * i + 2];
    }
    rtc::ClusterSet cs;
    rtc::build(v.data(), n_tris, center, radius, cs);
    rtv::VoxelSet vs;
    const double abs_extent = std::sqrt((double)center[0] * center[0] + (double)center[1] * center[1] +
                                        (double)center[2] * center[2]) + 3.0 * (double)radius;
    rtv::build(v.data(), n_tris, cs.rho, abs_extent, voxel_res, vs);
    const rtv::Grid &G = vs.grid;
    const long long ncell = (long long)G.nx * G.ny * G.nz;
    long long bad = 0, full = 0, fresh = 0;
    bad += (long long)vs.off.size() != ncell * rtv::CELL_STRIDE + 1 || vs.off[0] != 0u ||
           vs.off.back() != (uint32_t)vs.entries.size();
    auto list = [&](long long cell, int f) {
        const uint32_t a = vs.off[(size_t)cell * rtv::CELL_STRIDE + f], b = vs.off[(size_t)cell * rtv::CELL_STRIDE + f + 1];
        std::vector<int> out(vs.entries.begin() + a, vs.entries.begin() + b);
        return out;
    };
    const double csz = (double)G.cs;
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : bad, full, fresh)
    for (long long cell = 0; cell < ncell; cell++) {
        const int ix = (int)(cell % G.nx), iy = (int)((cell / G.nx) % G.ny), iz = (int)(cell / ((long long)G.nx * G.ny));
        bad += vs.off[(size_t)cell * rtv::CELL_STRIDE + 7] != vs.off[(size_t)(cell + 1) * rtv::CELL_STRIDE];
        for (int f = 0; f <= rtv::CELL_LISTS; f++)
            bad += vs.off[(size_t)cell * rtv::CELL_STRIDE + f] > vs.off[(size_t)cell * rtv::CELL_STRIDE + f + 1];
        std::vector<int> all = list(cell, rtv::LIST_START);
        full += (long long)all.size();
        std::vector<int> sorted_all = all;
        std::sort(sorted_all.begin(), sorted_all.end());
        bad += std::adjacent_find(sorted_all.begin(), sorted_all.end()) != sorted_all.end();
        // list 6 against the box / triangle test itself
        const double cc[3] = {(double)G.gminx + (ix + 0.5) * csz, (double)G.gminy + (iy + 0.5) * csz,
                              (double)G.gminz + (iz + 0.5) * csz};
        for (int t : sorted_all) bad += t < 0 || t >= n_tris || !(cs.rho[t] >= 0.0);
        if ((cell & 15) == 0) // (every 16th cell against every triangle: the test is the expensive part)
            for (int t = 0; t < n_tris; t++) {
                bool in = false;
                if (cs.rho[t] >= 0.0) {
                    const double g = cs.rho[t] + vs.margin;
                    const double hh[3] = {0.5 * csz + g, 0.5 * csz + g, 0.5 * csz + g};
                    double a[3], b[3], c[3];
                    for (int q = 0; q < 3; q++) {
                        a[q] = (double)v[3 * t].pos[q] - cc[q];
                        b[q] = (double)v[3 * t + 1].pos[q] - cc[q];
                        c[q] = (double)v[3 * t + 2].pos[q] - cc[q];
                    }
                    in = rtv::tri_box_overlap(a, b, c, hh);
                }
                bad += in != std::binary_search(sorted_all.begin(), sorted_all.end(), t);
            }
        for (int f = 0; f < 6; f++) {
            const int a = f >> 1, step = (f & 1) ? 1 : -1;
            const int jx = ix + (a == 0 ? step : 0), jy = iy + (a == 1 ? step : 0), jz = iz + (a == 2 ? step : 0);
            std::vector<int> theirs;
            if (jx >= 0 && jx < G.nx && jy >= 0 && jy < G.ny && jz >= 0 && jz < G.nz) {
                theirs = list(((long long)jz * G.ny + jy) * G.nx + jx, rtv::LIST_START);
                std::sort(theirs.begin(), theirs.end());
            }
            std::vector<int> want; // in the order of list 6
            for (int t : all)
                if (!std::binary_search(theirs.begin(), theirs.end(), t)) want.push_back(t);
            const std::vector<int> got = list(cell, f);
            fresh += (long long)got.size();
            bad += got != want;
        }
    }
    stats[0] = ncell; stats[1] = full; stats[2] = fresh; stats[3] = bad;
}
