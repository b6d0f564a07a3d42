// rt_voxel.h -- a uniform grid of cells in space for the closest-hit traversal of arbitrary rays
// (bounce passes): per cell, the list of triangles that come within rho of it; a ray walks the cells
// along its line front to back, tests the triangles listed there, and stops as soon as no cell
// further on can hold a hit that beats the one it has.
//
// Which pairs this covers.  rt_filter.h ("Plane-band test") splits the pairs
// the reference's arithmetic can accept in two: (B) ill-conditioned ones -- direction nearly in the
// triangle's plane and origin nearly on it -- which the band sweep finds, and (A) well-conditioned
// ones, |a^| >= K E, for which the point P where the ray's LINE meets the triangle's plane lies within
// rho of the triangle.  The walk finds every pair of kind (A) that can matter:
//   * the build lists triangle k in every cell that comes within rho_k + m of it (m: the margin that
//     covers the walk's rounding, below), so the cell that holds P lists k;
//   * P = o + t d with t >= -0.22 rho / |d| (a computed t^ > 0 forces that, see the bound on t^ below),
//     so a walk that starts 0.25 rho_max + m behind the origin and visits every cell the line crosses from
//     there on meets the cell of P;
//   * for a pair of kind (A) the computed t^ and the exact t are tied by
//         t^ >= 0.948 t - 0.0523 min(|e1|,|e2|) / |d|        (t >= 0)
//     [t^ = N^t / a^ (1 + 2 eps); |N^t - Nt| <= 8 eps |s||e1||e2| and |a^| >= K 9 eps |d| S max|e|
//     give |dNt / a^| <= 8/(9K) min|e| / |d| = 0.0523 min|e| / |d|; |a^ - a| <= 7/8 E <= 7/(8K) |a^|
//     gives t (1 - 0.0515)], and rho = 0.1275 (|e1| + |e2|) >= 0.255 min|e|, so
//         t^ >= 0.94 t - 0.25 rho_max / |d|   with room to spare.
//     A cell entered at parameter T therefore holds only hit points with t >= T, whose computed
//     t^ is at least 0.94 T - 0.25 rho_max/|d|: once that exceeds the best t^ found so far, nothing
//     further on can replace it (equal t^ does not stop the walk, so the lowest index among equal
//     t^ is still found), and the walk ends.
// Triangles that take no part (never accepted, or "always in the band": too large, slivers) are not
// listed; the band sweep tests those against every ray.  Rays outside the band records' range
// ("wild") do not walk: the band sweep has all their pairs.
//
// Rounding of the walk.  The walk keeps, per axis, the parameter at which the line leaves the
// current cell, computed from the cell's index each time: t_k = (gmin + i cs - o) / d, off from the
// exact crossing by a few ulp of its operands -- in position along that axis less than
// 3e-7 abs_extent, whatever the size of d.  At any parameter the walk's cell therefore differs
// from the exact one only along axes whose crossing lies within that error: the point is within
// m = 2^-18 abs_extent + 2^-12 cs of the walk's cell, which is what the build allows for.  An axis
// along which the ray moves less than m/4 over the whole box is not stepped at all.
//
// The walk compiles for the device (hipcc) and for the host (tests/native/filter_model.cpp); the build
// at the end is host code.
#pragma once

#include "rt_filter.h"

#include "../../include/rt.h"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <system_error>
#include <thread>
#include <vector>

namespace rtv {

struct Grid {         // (named members, not arrays: an axis chosen per lane must not become an indexed load)
    float gminx, gminy, gminz; // lower corner of cell (0, 0, 0)
    float gmaxx, gmaxy, gmaxz; // upper corner of the last cell
    float cs;        // edge length of the (cubic) cells
    float inv_cs;
    int nx, ny, nz;  // cells per axis
    float back;      // distance: the walk starts this far behind the ray's origin (0.25 rho_max + margin)
    float slack;     // distance: 0.25 rho_max, the absolute term of the bound on t^
    float flat;      // an axis with |d_axis| <= |d| * flat is not stepped
};

constexpr float STOP_FACTOR = 0.94f;
constexpr int LIST_ALIGN = 4; // entries a lane takes from a list at a time (one 16-byte load)
// Every cell has SEVEN lists.  List 6 holds every triangle that comes near the cell; list f (0..5) holds those
// of them that do NOT stand on list 6 of the face neighbour f as well (f = 2a: the neighbour one cell down
// along axis a, f = 2a + 1: one cell up).  A walk that enters a cell through a face has just been through the
// neighbour behind that face and has put that cell's whole list to the test, so in the new cell it reads the
// list of what is new there: by induction over the cells of the walk every triangle of every list 6 on the way
// is tested in the first cell of an unbroken run of cells that list it.  (A triangle's cells are those that
// meet a convex body, so such runs are rarely broken; a broken run costs a second test, never a missed one.)
// A walk reads list 6 in the cell it starts in.  On the 64K-triangle soup a list of new triangles holds 2
// where the whole list holds 5, so most cells cost a walk one step of four entries instead of two.
// off[8 cell + f] is where list f of the cell starts in `entries`, off[8 cell + f + 1] where it ends
// (off[8 cell + 7] = off[8 (cell + 1)]); the lists are not padded.
constexpr int CELL_LISTS = 7;
constexpr int CELL_STRIDE = 8;
constexpr int LIST_START = 6; // the list a walk reads in its first cell: everything
RT_HD float plane_at(const Grid &G, float gmin, int i) { return fmaf((float)i, G.cs, gmin); }

struct Walk {
    int ix, iy, iz;          // the current cell
    float tx, ty, tz;        // parameter at which the line leaves it along each axis (+inf: never)
    float invx, invy, invz;  // 1 / d per axis; its sign is the step's direction
    float t_enter;           // parameter at which the current cell was entered
    float inv_dn;            // 1 / |d|
    int face;                // the face neighbour the walk came from (0..5), LIST_START in its first cell
};

RT_HD int cell_index(const Grid &G, const Walk &w) { return (w.iz * G.ny + w.iy) * G.nx + w.ix; }
// the list of the cell the walk is in that holds what it has not tested yet
RT_HD int list_index(const Grid &G, const Walk &w) { return cell_index(G, w) * CELL_STRIDE + w.face; }

// one axis of the slab clip; false: the line misses the box
RT_HD bool clip_axis(float o, float d, float inv, bool flat, float lo, float hi, float &t0, float &t1)
{
    if (flat) return o >= lo && o <= hi;
    const float ta = (lo - o) * inv, tb = (hi - o) * inv;
    t0 = fmaxf(t0, fminf(ta, tb));
    t1 = fminf(t1, fmaxf(ta, tb));
    return true;
}

RT_HD float exit_param(const Grid &G, float gmin, int i, float o, float inv, bool flat)
{
    if (flat) return INFINITY;
    return (plane_at(G, gmin, inv > 0.0f ? i + 1 : i) - o) * inv;
}

RT_HD int start_index(float p, float gmin, float inv_cs, int n)
{
    const float q = floorf((p - gmin) * inv_cs);
    if (!(q > 0.0f)) return 0;
    if (q >= (float)n) return n - 1;
    return (int)q;
}

// Places the walk in the cell where the line enters the grid, or 0.25 rho_max + margin behind the ray's
// origin, or at parameter t_from (a walk taken up again: the parameter at which it had entered the
// cell it was in; -inf for a new walk), whichever comes last.  false: the line misses the grid, or
// the ray is not one the walk handles (zero or non-finite direction).
RT_HD bool walk_start(const Grid &G, float ox, float oy, float oz, float dx, float dy, float dz, Walk &w,
                      float t_from = -INFINITY)
{
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    if (!(dn > 0.0f) || !(dn < INFINITY)) return false;
    w.inv_dn = 1.0f / dn;
    const float lim = dn * G.flat;
    const bool fx = !(fabsf(dx) > lim), fy = !(fabsf(dy) > lim), fz = !(fabsf(dz) > lim);
    w.invx = 1.0f / dx; w.invy = 1.0f / dy; w.invz = 1.0f / dz;
    float t0 = fmaxf(-G.back * w.inv_dn, t_from), t1 = INFINITY;
    if (!clip_axis(ox, dx, w.invx, fx, G.gminx, G.gmaxx, t0, t1)) return false;
    if (!clip_axis(oy, dy, w.invy, fy, G.gminy, G.gmaxy, t0, t1)) return false;
    if (!clip_axis(oz, dz, w.invz, fz, G.gminz, G.gmaxz, t0, t1)) return false;
    if (!(t0 <= t1)) return false;
    w.face = LIST_START;
    w.t_enter = t0;
    w.ix = start_index(fmaf(t0, dx, ox), G.gminx, G.inv_cs, G.nx);
    w.iy = start_index(fmaf(t0, dy, oy), G.gminy, G.inv_cs, G.ny);
    w.iz = start_index(fmaf(t0, dz, oz), G.gminz, G.inv_cs, G.nz);
    w.tx = exit_param(G, G.gminx, w.ix, ox, w.invx, fx);
    w.ty = exit_param(G, G.gminy, w.iy, oy, w.invy, fy);
    w.tz = exit_param(G, G.gminz, w.iz, oz, w.invz, fz);
    return true;
}

// Moves to the next cell along the line.  false: the walk has left the grid.  (Written without
// branches on the axis: the lanes of a wavefront step along different axes.)
RT_HD bool walk_step(const Grid &G, float ox, float oy, float oz, Walk &w)
{
    // (values first: a conditional expression over members is a conditional lvalue, which turns the
    // struct into an indexed array in memory)
    const float tx = w.tx, ty = w.ty, tz = w.tz, invx = w.invx, invy = w.invy, invz = w.invz;
    const int ix = w.ix, iy = w.iy, iz = w.iz;
    const float gx = G.gminx, gy = G.gminy, gz = G.gminz;
    const int nx = G.nx, ny = G.ny, nz = G.nz;
    const bool sx = tx <= ty && tx <= tz;
    const bool sy = !sx && ty <= tz;
    const float inv = sx ? invx : (sy ? invy : invz);
    const float o = sx ? ox : (sy ? oy : oz);
    const float gm = sx ? gx : (sy ? gy : gz);
    const int nn = sx ? nx : (sy ? ny : nz);
    int i = sx ? ix : (sy ? iy : iz);
    w.t_enter = sx ? tx : (sy ? ty : tz);
    const bool up = inv > 0.0f;
    i += up ? 1 : -1;
    // (going up along an axis the cell left behind is the neighbour one cell down, and the other way round)
    w.face = (sx ? 0 : (sy ? 2 : 4)) + (up ? 0 : 1);
    // as exit_param() forms it
    const float tn = (fmaf((float)(up ? i + 1 : i), G.cs, gm) - o) * inv;
    w.ix = sx ? i : ix; w.tx = sx ? tn : tx;
    w.iy = sy ? i : iy; w.ty = sy ? tn : ty;
    w.iz = (sx || sy) ? iz : i; w.tz = (sx || sy) ? tz : tn;
    return i >= 0 && i < nn;
}

// true: no cell entered at parameter t_enter or later can hold a hit whose computed t^ is <= tbest
RT_HD bool walk_done_at(const Grid &G, float inv_dn, float t_enter, float tbest)
{
    return fmaf(STOP_FACTOR, t_enter, -(G.slack * inv_dn)) > tbest;
}
RT_HD bool walk_done(const Grid &G, const Walk &w, float tbest) { return walk_done_at(G, w.inv_dn, w.t_enter, tbest); }

// ---------------------------------------------------------------------------------------------
// The cells of a walk by rank.  walk_step() finds the cells one after the other; a wavefront that
// wants the next 64 cells of ONE walk at once (traverse_voxel_long_kernel) would have every lane repeat
// those 64 steps.  But the walk is a merge of three sorted sequences: along each axis the line leaves
// the cells i0, i0 +- 1, i0 +- 2, ... at parameters that depend on the cell's index alone,
//     E(m) = axis_exit(i0 +- m)          ("event" m of the axis: the step from i0 +- m to i0 +- (m + 1)),
// nondecreasing in m (every operation of axis_exit is monotone), and walk_step() always takes the
// smallest of the three pending parameters, x before y before z among equal ones.  Step number r of
// the walk is therefore the event of rank r in the order (parameter, axis, m), and the rank of event
// L of an axis is L + (events of the other two axes that come before it), two binary searches.  Lane
// L of a wavefront ranks event L of each axis; the event of rank r yields the walk's state after
// r + 1 steps.  tests/native/filter_model.cpp checks, on the host, that this reproduces walk_step()'s
// sequence state for state.
// ---------------------------------------------------------------------------------------------
RT_HD float axis_exit(const Grid &G, float gmin, int i, float o, float inv)
{
    return (fmaf((float)(inv > 0.0f ? i + 1 : i), G.cs, gmin) - o) * inv; // as exit_param() and walk_step() form it
}

struct AxisSeq {  // the events of one axis, counted from the walk's current cell
    float gmin, o, inv;
    int i0, n;    // current index, cells along the axis
    bool flat;    // the axis is not stepped (walk_start's criterion): its events never come
};

RT_HD int seq_index(const AxisSeq &s, int m) { return s.inv > 0.0f ? s.i0 + m : s.i0 - m; }
RT_HD float seq_t(const Grid &G, const AxisSeq &s, int m)
{
    return s.flat ? INFINITY : axis_exit(G, s.gmin, seq_index(s, m), s.o, s.inv);
}

// how many of the events 0..63 of s come before an event at parameter t of another axis; ties_first: an
// event of s at exactly t comes first (s's axis has priority)
RT_HD int seq_count_before(const Grid &G, const AxisSeq &s, float t, bool ties_first)
{
    int c = 0;
#define RT_SEQ_BEFORE(m) (seq_t(G, s, (m)) < t || (ties_first && seq_t(G, s, (m)) == t))
    for (int sh = 32; sh > 0; sh >>= 1)
        if (RT_SEQ_BEFORE(c + sh - 1)) c += sh;
    if (c == 63 && RT_SEQ_BEFORE(63)) c = 64;
#undef RT_SEQ_BEFORE
    return c;
}

// Event L of axis a (b, c: the other two axes; b_first / c_first: their events win ties against a's).
// Returns its rank; t: its parameter; cb, cc: the events of b and c before it.  After it the walk is in
// cell (seq_index(a, L + 1), seq_index(b, cb), seq_index(c, cc)), entered at t.
RT_HD int event_rank(const Grid &G, const AxisSeq &a, int L, const AxisSeq &b, bool b_first, const AxisSeq &c,
                     bool c_first, float &t, int &cb, int &cc)
{
    t = seq_t(G, a, L);
    cb = seq_count_before(G, b, t, b_first);
    cc = seq_count_before(G, c, t, c_first);
    return L + cb + cc;
}

// the three sequences of a walk in its current state (d: the ray's direction, as given to walk_start)
RT_HD void walk_sequences(const Grid &G, float ox, float oy, float oz, float dx, float dy, float dz, const Walk &w,
                          AxisSeq &sx, AxisSeq &sy, AxisSeq &sz)
{
    const float dn = sqrtf(dx * dx + dy * dy + dz * dz);
    const float lim = dn * G.flat;
    sx = AxisSeq{G.gminx, ox, w.invx, w.ix, G.nx, !(fabsf(dx) > lim)};
    sy = AxisSeq{G.gminy, oy, w.invy, w.iy, G.ny, !(fabsf(dy) > lim)};
    sz = AxisSeq{G.gminz, oz, w.invz, w.iz, G.nz, !(fabsf(dz) > lim)};
}

// the event of an axis that leaves the grid: 64 or more when it is not among the next 64
RT_HD int seq_exit_event(const AxisSeq &s)
{
    if (s.flat) return 1 << 20;
    return s.inv > 0.0f ? s.n - 1 - s.i0 : s.i0;
}

// ---------------------------------------------------------------------------------------------
// Host-side build.
// ---------------------------------------------------------------------------------------------
struct VoxelSet {
    Grid grid{};
    std::vector<uint32_t> off;     // CELL_STRIDE * cells + 1
    std::vector<int32_t> entries;  // triangle indices: cell after cell, list after list
    double rho_max = 0.0;
    double margin = 0.0;
};

// Does the triangle (a, b, c), given relative to the box's centre, come within the box of half
// sizes h?  Separating axes: the three box normals, the triangle's normal, the nine cross products.
inline bool tri_box_overlap(const double a[3], const double b[3], const double c[3], const double h[3])
{
    for (int k = 0; k < 3; k++) {
        const double lo = std::min(a[k], std::min(b[k], c[k])), hi = std::max(a[k], std::max(b[k], c[k]));
        if (lo > h[k] || hi < -h[k]) return false;
    }
    const double e[3][3] = {{b[0] - a[0], b[1] - a[1], b[2] - a[2]},
                            {c[0] - b[0], c[1] - b[1], c[2] - b[2]},
                            {a[0] - c[0], a[1] - c[1], a[2] - c[2]}};
    const double *v[3] = {a, b, c};
    for (int i = 0; i < 3; i++)       // edge i
        for (int k = 0; k < 3; k++) { // box axis k: axis = unit_k x e_i
            const int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            const double ax1 = -e[i][k2], ax2 = e[i][k1]; // components along k1, k2
            double lo = 0.0, hi = 0.0;
            for (int q = 0; q < 3; q++) {
                const double p = ax1 * v[q][k1] + ax2 * v[q][k2];
                if (q == 0 || p < lo) lo = p;
                if (q == 0 || p > hi) hi = p;
            }
            const double r = std::fabs(ax1) * h[k1] + std::fabs(ax2) * h[k2];
            if (lo > r || hi < -r) return false;
        }
    const double n[3] = {e[0][1] * e[1][2] - e[0][2] * e[1][1], e[0][2] * e[1][0] - e[0][0] * e[1][2],
                         e[0][0] * e[1][1] - e[0][1] * e[1][0]};
    const double dist = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
    const double r = std::fabs(n[0]) * h[0] + std::fabs(n[1]) * h[1] + std::fabs(n[2]) * h[2];
    return std::fabs(dist) <= r;
}

// rho[t] >= 0 marks the triangles that take part (the ones the band records leave to the spatial
// sweep); res = cells along the longest axis of their bounding box, 0 = chosen from their number.
// abs_extent bounds the magnitude of every coordinate involved (origins within the records' range
// included).
inline void build(const rt_vertex *v, int num_tris, const std::vector<double> &rho, double abs_extent, int res,
                  VoxelSet &out)
{
    out = VoxelSet{};
    Grid &G = out.grid;
    float gmin[3] = {0.0f, 0.0f, 0.0f}, gmax[3] = {0.0f, 0.0f, 0.0f};
    int gn[3] = {1, 1, 1};
    auto store = [&]() {
        G.gminx = gmin[0]; G.gminy = gmin[1]; G.gminz = gmin[2];
        G.gmaxx = gmax[0]; G.gmaxy = gmax[1]; G.gmaxz = gmax[2];
        G.nx = gn[0]; G.ny = gn[1]; G.nz = gn[2];
    };
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    size_t count = 0;
    for (int t = 0; t < num_tris; t++) {
        if (!(rho[t] >= 0.0)) continue;
        out.rho_max = std::max(out.rho_max, rho[t]);
        for (int k = 0; k < 3; k++)
            for (int a = 0; a < 3; a++) {
                const double x = v[3 * t + k].pos[a];
                if (count == 0 || x < lo[a]) lo[a] = x;
                if (count == 0 || x > hi[a]) hi[a] = x;
            }
        count++;
    }
    if (count == 0) { // an empty grid: one cell, no entries
        G.cs = 1.0f; G.inv_cs = 1.0f; G.back = 0.0f; G.slack = 0.0f; G.flat = 0.0f;
        gmin[0] = 1.0f; // gmin > gmax: every line misses
        store();
        out.off.assign(CELL_STRIDE + 1, 0u);
        return;
    }
    double longest = 0.0;
    for (int a = 0; a < 3; a++) longest = std::max(longest, hi[a] - lo[a]);
    if (res <= 0) res = (int)std::lround(std::cbrt(4.0 * (double)count));
    res = std::max(1, std::min(res, 128));
    double margin0 = abs_extent * 3.814697265625e-06; // 2^-18
    double cs = std::max(longest / res, 64.0 * margin0);
    if (!(cs > 0.0)) cs = 1.0;
    G.cs = (float)cs;
    cs = (double)G.cs;
    G.inv_cs = (float)(1.0 / cs);
    out.margin = margin0 + cs * 0.000244140625; // 2^-12
    const double pad = out.rho_max + 4.0 * out.margin;
    size_t ncell = 1;
    for (int a = 0; a < 3; a++) {
        gmin[a] = (float)(lo[a] - pad);
        if ((double)gmin[a] > lo[a] - pad) gmin[a] = std::nextafterf(gmin[a], -INFINITY);
        const double ext = hi[a] + pad - (double)gmin[a];
        int n = (int)std::ceil(ext / cs) + 1;
        n = std::max(1, std::min(n, 2 * 128 + 8));
        gn[a] = n;
        gmax[a] = plane_at(G, gmin[a], n);
        ncell *= (size_t)n;
    }
    store();
    double diag = 0.0;
    for (int a = 0; a < 3; a++) diag += ((double)gmax[a] - gmin[a]) * ((double)gmax[a] - gmin[a]);
    diag = std::sqrt(diag) + pad;
    G.back = (float)(0.25 * out.rho_max * 1.001 + out.margin);
    if ((double)G.back < 0.25 * out.rho_max + out.margin) G.back = std::nextafterf(G.back, INFINITY);
    G.slack = (float)(0.25 * out.rho_max * 1.001);
    if ((double)G.slack < 0.25 * out.rho_max) G.slack = std::nextafterf(G.slack, INFINITY);
    G.flat = (float)(0.25 * out.margin / diag);

    // (cell, triangle) pairs, then a counting sort by cell
    std::vector<uint32_t> cells;
    std::vector<int32_t> tris;
    std::vector<unsigned char> absent; // per pair: bit f set = the triangle is not on the list of face neighbour f
    std::vector<uint32_t> cnt(ncell * CELL_STRIDE + 1, 0u);
    // The box/triangle tests are most of a build (tens of millions on a large mesh): the triangles are cut into
    // runs, a host thread per run files its pairs, and the runs are joined in order, so the lists come out the
    // same whatever the number of threads.
    struct Part {
        std::vector<uint32_t> cells;
        std::vector<int32_t> tris;
        std::vector<unsigned char> absent;
    };
    unsigned int n_threads = std::thread::hardware_concurrency();
    n_threads = std::max(1u, std::min(n_threads, 16u));
    if (num_tris < 8192) n_threads = 1;
    std::vector<Part> parts(n_threads);
    auto file_run = [&](unsigned int part) {
        Part &P = parts[part];
        const int t_begin = (int)((long long)num_tris * part / n_threads);
        const int t_end = (int)((long long)num_tris * (part + 1) / n_threads);
        std::vector<uint32_t> mine; // the cells of one triangle
        std::vector<int> mine_xyz;
        std::vector<unsigned char> in_box; // per cell of the triangle's box of cells, one cell of rim around it: listed?
        for (int t = t_begin; t < t_end; t++) {
            if (!(rho[t] >= 0.0)) continue;
            const double g = rho[t] + out.margin;
            int c0[3], c1[3];
            for (int a = 0; a < 3; a++) {
                double l = v[3 * t].pos[a], h = l;
                for (int k = 1; k < 3; k++) {
                    l = std::min(l, (double)v[3 * t + k].pos[a]);
                    h = std::max(h, (double)v[3 * t + k].pos[a]);
                }
                c0[a] = std::max(0, std::min(gn[a] - 1, (int)std::floor((l - g - (double)gmin[a]) / cs)));
                c1[a] = std::max(0, std::min(gn[a] - 1, (int)std::floor((h + g - (double)gmin[a]) / cs)));
            }
            const double hh[3] = {0.5 * cs + g, 0.5 * cs + g, 0.5 * cs + g};
            mine.clear();
            mine_xyz.clear();
            const int bx = c1[0] - c0[0] + 3, by = c1[1] - c0[1] + 3, bz = c1[2] - c0[2] + 3;
            in_box.assign((size_t)bx * by * bz, 0);
            auto box_at = [&](int ix, int iy, int iz) -> unsigned char & {
                return in_box[((size_t)(iz - c0[2] + 1) * by + (iy - c0[1] + 1)) * bx + (ix - c0[0] + 1)];
            };
            for (int iz = c0[2]; iz <= c1[2]; iz++)
                for (int iy = c0[1]; iy <= c1[1]; iy++)
                    for (int ix = c0[0]; ix <= c1[0]; ix++) {
                        const double cc[3] = {(double)gmin[0] + (ix + 0.5) * cs, (double)gmin[1] + (iy + 0.5) * cs,
                                              (double)gmin[2] + (iz + 0.5) * cs};
                        double a[3], b[3], c[3];
                        for (int q = 0; q < 3; q++) {
                            a[q] = (double)v[3 * t].pos[q] - cc[q];
                            b[q] = (double)v[3 * t + 1].pos[q] - cc[q];
                            c[q] = (double)v[3 * t + 2].pos[q] - cc[q];
                        }
                        if (!tri_box_overlap(a, b, c, hh)) continue;
                        const uint32_t cell = (uint32_t)(((size_t)iz * gn[1] + iy) * gn[0] + ix);
                        mine.push_back(cell);
                        mine_xyz.push_back(ix); mine_xyz.push_back(iy); mine_xyz.push_back(iz);
                        box_at(ix, iy, iz) = 1;
                    }
            for (size_t j = 0; j < mine.size(); j++) {
                const int ix = mine_xyz[3 * j], iy = mine_xyz[3 * j + 1], iz = mine_xyz[3 * j + 2];
                // bit f: the face neighbour f lists the triangle too (cells outside the grid are in the rim: not listed)
                const int here = box_at(ix - 1, iy, iz) | box_at(ix + 1, iy, iz) << 1 | box_at(ix, iy - 1, iz) << 2 |
                                 box_at(ix, iy + 1, iz) << 3 | box_at(ix, iy, iz - 1) << 4 | box_at(ix, iy, iz + 1) << 5;
                P.cells.push_back(mine[j]);
                P.tris.push_back(t);
                P.absent.push_back((unsigned char)(~here & 63));
            }
        }
    };
    if (n_threads == 1) {
        file_run(0);
    } else {
        std::vector<std::thread> pool;
        unsigned int started = 0;
        try {
            for (; started < n_threads; started++) pool.emplace_back(file_run, started);
        } catch (const std::system_error &) {
            // (no more threads to be had: this one files the runs that are left)
        }
        for (unsigned int part = started; part < n_threads; part++) file_run(part);
        for (std::thread &th : pool) th.join();
    }
    for (const Part &P : parts) {
        for (size_t i = 0; i < P.cells.size(); i++) {
            const size_t at = (size_t)P.cells[i] * CELL_STRIDE;
            for (int f = 0; f < 6; f++)
                if ((P.absent[i] >> f) & 1) cnt[at + f + 1]++;
            cnt[at + LIST_START + 1]++;
        }
        cells.insert(cells.end(), P.cells.begin(), P.cells.end());
        tris.insert(tris.end(), P.tris.begin(), P.tris.end());
        absent.insert(absent.end(), P.absent.begin(), P.absent.end());
    }
    // counts -> positions (slot 8 cell + 7 holds nothing of its own: it is where the next cell's lists start)
    for (size_t i = 0; i < ncell * CELL_STRIDE; i++) cnt[i + 1] += cnt[i];
    out.off = cnt;
    out.entries.assign(cnt[ncell * CELL_STRIDE], -1);
    std::vector<uint32_t> cur(cnt.begin(), cnt.end() - 1);
    for (size_t i = 0; i < cells.size(); i++) {
        const size_t at = (size_t)cells[i] * CELL_STRIDE;
        for (int f = 0; f < 6; f++)
            if ((absent[i] >> f) & 1) out.entries[cur[at + f]++] = tris[i];
        out.entries[cur[at + LIST_START]++] = tris[i];
    }
}
} // namespace rtv
