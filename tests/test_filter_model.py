"""Randomized check that the traversal filter never rejects a (ray, triangle) pair
that the oracle's Moller-Trumbore accepts.  Runs the filter header the kernels use,
compiled for the host (tests/native/filter_model.cpp), against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


@pytest.fixture(scope="module")
def fm():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "..", "oracle")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "native")])
    lib = C.CDLL(os.path.join(HERE, "native", "libfilter_model.so"))
    P = C.c_void_p
    lib.fm_check_general.argtypes = [C.c_int, P, C.c_int, P, P, P, C.c_float, P]
    lib.fm_check_origin.argtypes = [C.c_int, P, C.c_int, P, P, P, P, C.c_float, P]
    lib.fm_check_cone.argtypes = [C.c_int, P, C.c_int, P, P, P, P, P, C.c_float, C.c_float, P]
    return lib


def bounding_sphere(verts):
    v = verts.reshape(-1, 3).astype(np.float64)
    v = v[np.isfinite(v).all(1)]  # as rt_set_mesh does: such triangles can never be hit
    c = (v.min(0) + v.max(0)) / 2
    r = np.sqrt(((v - c) ** 2).sum(1).max()) * 1.0001 + 1e-30
    return c.astype(f32), float(r)


def tris_random(rs, n, extent, size, offset=(0, 0, 0)):
    c = rs.uniform(-extent, extent, size=(n, 1, 3)) + np.asarray(offset)
    return (c + rs.uniform(-size, size, size=(n, 3, 3))).astype(f32)


def tris_nasty(rs, n, extent):
    """needles, slivers, huge and tiny triangles mixed"""
    t = tris_random(rs, n, extent, 0.1).astype(np.float64)
    kind = rs.randint(0, 5, size=n)
    for i in range(n):
        if kind[i] == 0:    # needle: third vertex almost on the first edge
            a = rs.uniform(0, 1)
            t[i, 2] = t[i, 0] + a * (t[i, 1] - t[i, 0]) + rs.uniform(-1e-6, 1e-6, 3)
        elif kind[i] == 1:  # tiny
            t[i, 1:] = t[i, 0] + rs.uniform(-1e-4, 1e-4, (2, 3))
        elif kind[i] == 2:  # huge
            t[i, 1:] = t[i, 0] + rs.uniform(-50, 50, (2, 3))
        elif kind[i] == 3:  # exactly degenerate
            t[i, 2] = t[i, 0]
    return t.astype(f32)


def rays_through_features(rs, tris, origins, per_tri=6):
    """Directions aimed at points on (and a hair beside) triangle edges and vertices."""
    n = tris.shape[0]
    dirs, orgs = [], []
    for j in range(per_tri):
        w = rs.dirichlet([1, 1, 1], size=n)
        mode = j % 3
        if mode == 0:
            w[:, rs.randint(0, 3)] = 0.0         # on an edge
        elif mode == 1:
            w[:] = 0.0
            w[np.arange(n), rs.randint(0, 3, n)] = 1.0  # at a vertex
        w /= w.sum(1, keepdims=True)
        p = (w[:, :, None] * tris.astype(np.float64)).sum(1)
        p += rs.uniform(-1, 1, p.shape) * 10.0 ** rs.uniform(-9, -5, (n, 1))
        o = origins[rs.randint(0, origins.shape[0], n)]
        d = p - o
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-300
        dirs.append(d)
        orgs.append(o)
    return np.concatenate(orgs).astype(f32), np.concatenate(dirs).astype(f32)


def run_general(fm, tris, org, dirs):
    c, r = bounding_sphere(tris)
    stats = (C.c_longlong * 5)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    fm.fm_check_general(tris.shape[0], tris.ctypes.data, org.shape[0], org.ctypes.data, dirs.ctypes.data,
                        c.ctypes.data, r, stats)
    run_general.moment_passes = stats[4]  # the moment form of the filter: its pass count, for the selectivity test
    return list(stats)[:4]


def run_origin(fm, tris, O, org, dirs):
    c, r = bounding_sphere(tris)
    stats = (C.c_longlong * 4)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    O = np.ascontiguousarray(O, dtype=f32)
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    fm.fm_check_origin(tris.shape[0], tris.ctypes.data, org.shape[0], O.ctypes.data, org.ctypes.data,
                       dirs.ctypes.data, c.ctypes.data, r, stats)
    return list(stats)


SCENES = [
    dict(name="soup", extent=1.0, size=0.05, offset=(0, 0, 0)),
    dict(name="big_tris", extent=1.0, size=1.5, offset=(0, 0, 0)),
    dict(name="far_from_world_origin", extent=1.0, size=0.05, offset=(900.0, -400.0, 250.0)),
    dict(name="tiny_scene", extent=1e-3, size=1e-4, offset=(0, 0, 0)),
    dict(name="large_scene", extent=1e4, size=300.0, offset=(0, 0, 0)),
]


@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_general_form_is_conservative(fm, sc):
    rs = np.random.RandomState(11)
    tris = tris_random(rs, 1500, sc["extent"], sc["size"], sc["offset"])
    origins = (rs.uniform(-3, 3, (64, 3)) * sc["extent"] + np.asarray(sc["offset"]))
    org, dirs = rays_through_features(rs, tris, origins)
    # as intersect does: origin + dir*0.001 (float32)
    org = (org + dirs * f32(0.001)).astype(f32)
    pairs, passed, acc, bad = run_general(fm, tris, org, dirs)
    assert bad == 0
    assert acc > 1000               # the feature-aimed rays do exercise accepted pairs
    assert passed < 0.2 * pairs     # and the filter does reject most pairs
    assert run_general.moment_passes < 0.2 * pairs  # so does its moment form


def test_general_form_nasty_triangles(fm):
    rs = np.random.RandomState(5)
    tris = tris_nasty(rs, 3000, 2.0)
    origins = rs.uniform(-6, 6, (64, 3))
    org, dirs = rays_through_features(rs, tris, origins)
    org = (org + dirs * f32(0.001)).astype(f32)
    pairs, passed, acc, bad = run_general(fm, tris, org, dirs)
    assert bad == 0 and acc > 1000


def test_general_form_rays_starting_on_triangles(fm):
    """bounce rays: origins on a surface, directions random and grazing"""
    rs = np.random.RandomState(6)
    tris = tris_random(rs, 2000, 1.0, 0.2)
    w = rs.dirichlet([1, 1, 1], size=2000)
    p = (w[:, :, None] * tris.astype(np.float64)).sum(1)
    d = rs.normal(size=(2000, 3))
    # half of them nearly in the plane of their triangle
    nrm = np.cross(tris[:, 1] - tris[:, 0], tris[:, 2] - tris[:, 0]).astype(np.float64)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d[::2] -= (d[::2] * nrm[::2]).sum(1, keepdims=True) * nrm[::2] * (1 - 1e-4)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    org = (p.astype(f32) + d.astype(f32) * f32(0.001)).astype(f32)
    pairs, passed, acc, bad = run_general(fm, tris, org, d.astype(f32))
    assert bad == 0 and acc > 1000


@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_shared_origin_primary_rays(fm, sc):
    """raygen rays: one origin O, reference origin = O + dir*0.001"""
    rs = np.random.RandomState(21)
    tris = tris_random(rs, 1500, sc["extent"], sc["size"], sc["offset"])
    for trial in range(3):
        O = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
        _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64))
        org = (O[None, :] + dirs * f32(0.001)).astype(f32)
        pairs, passed, acc, bad = run_origin(fm, tris, O, org, dirs)
        assert bad == 0
        assert acc > 1000 and passed < 0.2 * pairs


@pytest.mark.parametrize("sc", SCENES[:3], ids=[s["name"] for s in SCENES[:3]])
def test_shared_origin_shadow_rays(fm, sc):
    """shadow rays as color builds them: from surface points p toward the light L with
    lightDir = normalize(L - p), origin p + lightDir*1e-7 (raycol_cs.glsl:39-41)"""
    rs = np.random.RandomState(31)
    tris = tris_random(rs, 1500, sc["extent"], sc["size"], sc["offset"])
    L = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
    # choose p so that the segment p -> L grazes triangle features: aim from L through a
    # feature point and continue to a random distance behind it
    _, dirs = rays_through_features(rs, tris, L[None, :].astype(np.float64))
    dist = rs.uniform(0.5, 5.0, (dirs.shape[0], 1)) * sc["extent"]
    p = (L[None, :].astype(np.float64) + dirs.astype(np.float64) * dist).astype(f32)
    dlt = (L[None, :] - p).astype(f32)
    inv = (f32(1.0) / np.sqrt((dlt * dlt).sum(1, dtype=f32))).astype(f32)
    ld = (dlt * inv[:, None]).astype(f32)
    org = (p + ld * f32(0.0000001)).astype(f32)
    pairs, passed, acc, bad = run_origin(fm, tris, L, org, ld)
    assert bad == 0
    assert acc > 1000 and passed < 0.2 * pairs


def test_degenerate_and_nonfinite_inputs_never_pass_wrongly(fm):
    rs = np.random.RandomState(41)
    tris = tris_random(rs, 64, 1.0, 0.3)
    tris[0, 1] = tris[0, 0]                 # zero edge
    tris[1, 2, 0] = np.nan                  # NaN vertex
    tris[2, 1, 1] = np.inf                  # infinite vertex
    org = rs.uniform(-3, 3, (256, 3)).astype(f32)
    dirs = rs.normal(size=(256, 3)).astype(f32)
    dirs[0] = 0.0                           # zero direction
    dirs[1, 0] = np.nan
    org[2, 1] = np.inf
    org[3] = np.nan
    pairs, passed, acc, bad = run_general(fm, tris, org, dirs)
    assert bad == 0
    O = f32([0.1, 0.2, -3.0])
    pairs, passed, acc, bad = run_origin(fm, tris, O, (O[None, :] + dirs * f32(0.001)).astype(f32), dirs)
    assert bad == 0


# -- cone pre-filter (level 1 of the shared-origin traversal) -------------------------------
def run_cone(fm, tris, O, org, dirs, sign, tmax=None):
    c, r = bounding_sphere(tris)
    stats = (C.c_longlong * 7)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    O = np.ascontiguousarray(O, dtype=f32)
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    tm = None if tmax is None else np.ascontiguousarray(tmax, dtype=f32)
    fm.fm_check_cone(tris.shape[0], tris.ctypes.data, org.shape[0], O.ctypes.data, org.ctypes.data,
                     dirs.ctypes.data, tm.ctypes.data if tm is not None else None, c.ctypes.data, r, sign, stats)
    return dict(zip(["pairs", "cone", "both", "accepts", "bad", "always", "wild"], list(stats)))


@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_cone_cascade_primary_rays(fm, sc):
    rs = np.random.RandomState(51)
    tris = tris_random(rs, 1500, sc["extent"], sc["size"], sc["offset"])
    for trial in range(3):
        O = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
        _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64))
        org = (O[None, :] + dirs * f32(0.001)).astype(f32)
        st = run_cone(fm, tris, O, org, dirs, 1.0)
        assert st["bad"] == 0, st
        assert st["accepts"] > 1000
        assert st["wild"] == 0
        assert st["always"] < 0.05 * tris.shape[0], st    # the cone is the norm, not the exception
        # and it rejects almost everything (rays here are all aimed at triangle features;
        # triangles as large as the scene, or float32 coordinates ~1000x the scene size, widen it)
        limit = {"big_tris": 0.6, "far_from_world_origin": 0.3}.get(sc["name"], 0.05)
        assert st["cone"] < limit * st["pairs"], st


def test_cone_cascade_origin_inside_the_scene_and_nasty_triangles(fm):
    rs = np.random.RandomState(52)
    tris = tris_nasty(rs, 3000, 2.0)
    for O in ([0.1, -0.2, 0.3], [0.0, 0.0, 0.0], [5.0, 1.0, -2.0]):
        O = f32(O)
        _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64))
        org = (O[None, :] + dirs * f32(0.001)).astype(f32)
        st = run_cone(fm, tris, O, org, dirs, 1.0)
        assert st["bad"] == 0 and st["accepts"] > 1000, st


def test_cone_cascade_origin_on_triangle_planes(fm):
    """The camera lies exactly in the plane of many triangles (a floor it stands on), on an edge
    line of some and at a vertex of one: those cones must fall back to 'always pass'."""
    rs = np.random.RandomState(53)
    n = 400
    xy = rs.uniform(-2, 2, (n, 3, 2))
    floor = np.concatenate([xy, np.zeros((n, 3, 1))], axis=2).astype(f32)     # all in z = 0
    O = f32([0.25, -0.5, 0.0])
    floor[0, 0] = O                                                               # a vertex at the origin
    floor[1, 0] = O + f32([1, 0, 0]); floor[1, 1] = O - f32([1, 0, 0])            # origin on an edge
    other = tris_random(rs, 600, 1.5, 0.3)
    tris = np.concatenate([floor, other])
    _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64))
    org = (O[None, :] + dirs * f32(0.001)).astype(f32)
    st = run_cone(fm, tris, O, org, dirs, 1.0)
    assert st["bad"] == 0, st
    assert st["always"] >= n


@pytest.mark.parametrize("sc", SCENES[:3], ids=[s["name"] for s in SCENES[:3]])
def test_cone_cascade_shadow_rays(fm, sc):
    rs = np.random.RandomState(54)
    tris = tris_random(rs, 1500, sc["extent"], sc["size"], sc["offset"])
    L = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
    _, dirs = rays_through_features(rs, tris, L[None, :].astype(np.float64))
    # surface points behind the aimed-at feature, at a random distance from the light; half of
    # them barely behind it (the occluder sits right at the ray's start) or barely in front
    dist = rs.uniform(0.05, 5.0, (dirs.shape[0], 1)) * sc["extent"]
    p = (L[None, :].astype(np.float64) + dirs.astype(np.float64) * dist).astype(f32)
    dlt = (L[None, :] - p).astype(f32)
    inv = (f32(1.0) / np.sqrt((dlt * dlt).sum(1, dtype=f32))).astype(f32)
    ld = (dlt * inv[:, None]).astype(f32)
    org = (p + ld * f32(0.0000001)).astype(f32)
    pm = (p - ld * f32(0.0000001)).astype(f32)
    tmax = np.sqrt(((L[None, :] - pm) ** 2).sum(1, dtype=f32)).astype(f32)
    st = run_cone(fm, tris, L, org, ld, -1.0, tmax)
    assert st["bad"] == 0, st
    assert st["accepts"] > 300, st
    limit = {"big_tris": 0.6, "far_from_world_origin": 0.3}.get(sc["name"], 0.05)
    assert st["cone"] < limit * st["pairs"], st


def test_cone_cascade_shadow_rays_light_near_surfaces(fm):
    """Occluders hugging the light: triangles whose plane passes within rounding of the light."""
    rs = np.random.RandomState(55)
    L = f32([0.3, 0.1, -0.2])
    tris = tris_random(rs, 800, 1.0, 0.3)
    for i in range(200):   # planes through (or a hair off) the light
        a, b = rs.normal(size=3), rs.normal(size=3)
        off = rs.normal(size=3) * 10.0 ** rs.uniform(-9, -4)
        c = rs.uniform(-1, 1, 2)
        tris[i] = (L + off + np.array([c[0] * a + c[1] * b + 0.3 * rs.normal() * a,
                                       c[0] * a + c[1] * b + 0.3 * rs.normal() * b,
                                       c[0] * a + c[1] * b - 0.3 * rs.normal() * (a + b)])).astype(f32)
    _, dirs = rays_through_features(rs, tris, L[None, :].astype(np.float64))
    dist = rs.uniform(0.05, 3.0, (dirs.shape[0], 1))
    p = (L[None, :].astype(np.float64) + dirs.astype(np.float64) * dist).astype(f32)
    dlt = (L[None, :] - p).astype(f32)
    inv = (f32(1.0) / np.sqrt((dlt * dlt).sum(1, dtype=f32))).astype(f32)
    ld = (dlt * inv[:, None]).astype(f32)
    org = (p + ld * f32(0.0000001)).astype(f32)
    pm = (p - ld * f32(0.0000001)).astype(f32)
    tmax = np.sqrt(((L[None, :] - pm) ** 2).sum(1, dtype=f32)).astype(f32)
    st = run_cone(fm, tris, L, org, ld, -1.0, tmax)
    assert st["bad"] == 0, st


def test_cone_cascade_wild_rays_fall_back_to_the_edge_filter(fm):
    """Shadow rays that start far outside the scene bound are outside the cones' validity."""
    rs = np.random.RandomState(56)
    tris = tris_random(rs, 500, 1.0, 0.2)
    L = f32([0.5, 2.0, -1.0])
    _, dirs = rays_through_features(rs, tris, L[None, :].astype(np.float64), per_tri=3)
    p = (L[None, :].astype(np.float64) + dirs.astype(np.float64) * 500.0).astype(f32)
    dlt = (L[None, :] - p).astype(f32)
    inv = (f32(1.0) / np.sqrt((dlt * dlt).sum(1, dtype=f32))).astype(f32)
    ld = (dlt * inv[:, None]).astype(f32)
    org = (p + ld * f32(0.0000001)).astype(f32)
    st = run_cone(fm, tris, L, org, ld, -1.0, np.full(ld.shape[0], 1e9, dtype=f32))
    assert st["bad"] == 0, st
    assert st["wild"] == ld.shape[0]


# -- clustered traversal of arbitrary rays (normal groups + band/slab, spatial clusters) ---------
CL_KEYS = ["pairs", "group", "member", "sphere", "to_exact", "accepts", "bad", "wild", "unclustered", "never",
           "bad_closest", "cells", "tests", "hits"]


def run_clustered(fm, tris, org, dirs, sphere=None, voxel_res=0):
    """`bad` counts pairs the oracle accepts that never reach the reference arithmetic (the walk over the
    cells in space not stopping early); `bad_closest` counts rays whose closest hit differs from the
    oracle's when the walk stops by its rule.  Both must be 0 (folded into "bad" for the callers)."""
    c, r = bounding_sphere(tris) if sphere is None else sphere
    stats = (C.c_longlong * 14)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    fm.fm_check_clustered.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_int, C.c_void_p]
    fm.fm_check_clustered(tris.shape[0], tris.ctypes.data, org.shape[0], org.ctypes.data, dirs.ctypes.data,
                          c.ctypes.data, r, voxel_res, stats)
    st = dict(zip(CL_KEYS, list(stats)))
    st["bad_pairs"] = st["bad"]
    st["bad"] = st["bad"] + st["bad_closest"]
    return st


def bounce_like_rays(rs, tris, n, extent, offset):
    """Origins on triangles, directions random -- what a bounce pass traces."""
    k = rs.randint(0, tris.shape[0], n)
    w = rs.dirichlet([1, 1, 1], size=n)
    p = (w[:, :, None] * tris[k].astype(np.float64)).sum(1)
    d = rs.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return (p + d * 0.001).astype(f32), d.astype(f32)


def in_plane_rays(rs, tris, n, spread):
    """Rays lying in (or tilted out of, by 1e-9 .. 1e-2 rad) the plane of a triangle, starting up to
    `spread` edge lengths away from it: where the reference's arithmetic accepts from rounding noise."""
    k = rs.randint(0, tris.shape[0], n)
    t = tris[k].astype(np.float64)
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    nrm = np.cross(e1, e2)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True) + 1e-300
    a, b = rs.uniform(-spread, spread, (n, 1)), rs.uniform(-spread, spread, (n, 1))
    start = t[:, 0] + a * e1 + b * e2
    # aim at a point of the triangle (half of them) or anywhere in the plane
    w = rs.dirichlet([1, 1, 1], size=n)
    target = (w[:, :, None] * t).sum(1)
    anywhere = t[:, 0] + rs.uniform(-spread, spread, (n, 1)) * e1 + rs.uniform(-spread, spread, (n, 1)) * e2
    target[1::2] = anywhere[1::2]
    d = target - start
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-300
    tilt = 10.0 ** rs.uniform(-9, -2, (n, 1)) * rs.choice([-1, 0, 1], size=(n, 1))
    d = d + nrm * tilt
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-300
    lift = 10.0 ** rs.uniform(-9, -3, (n, 1)) * rs.choice([-1, 0, 1], size=(n, 1))
    start = start + nrm * lift * np.linalg.norm(e1, axis=1, keepdims=True)
    return start.astype(f32), d.astype(f32)


@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_clustered_path_is_conservative(fm, sc):
    rs = np.random.RandomState(71)
    tris = tris_random(rs, 1500, sc["extent"], sc["size"], sc["offset"])
    origins = (rs.uniform(-1.5, 1.5, (64, 3)) * sc["extent"] + np.asarray(sc["offset"]))
    o1, d1 = rays_through_features(rs, tris, origins, per_tri=3)
    o1 = (o1 + d1 * f32(0.001)).astype(f32)
    o2, d2 = bounce_like_rays(rs, tris, 3000, sc["extent"], sc["offset"])
    o3, d3 = in_plane_rays(rs, tris, 3000, 4.0)
    st = run_clustered(fm, tris, np.concatenate([o1, o2, o3]), np.concatenate([d1, d2, d3]))
    assert st["bad"] == 0, st
    assert st["accepts"] > 1000, st
    if sc["name"] in ("soup", "tiny_scene", "large_scene"):
        # small triangles in a large scene: the two sweeps leave little for the edge filter
        assert st["unclustered"] < 0.02 * tris.shape[0], st
        assert st["member"] < 0.02 * st["pairs"] and st["sphere"] < 0.3 * st["pairs"], st


def test_clustered_path_in_plane_rays_far_from_their_triangle(fm):
    """Coplanar rays up to 30 edge lengths away: the band + slab must keep every pair the
    reference accepts out of rounding noise (there are such pairs: accepts counts them)."""
    rs = np.random.RandomState(72)
    tris = tris_random(rs, 800, 1.0, 0.1)
    o, d = in_plane_rays(rs, tris, 20000, 30.0)
    st = run_clustered(fm, tris, o, d)
    assert st["bad"] == 0, st
    assert st["accepts"] > 3000, st


def test_clustered_path_nasty_triangles_and_coplanar_floor(fm):
    rs = np.random.RandomState(73)
    n = 600
    xy = rs.uniform(-2, 2, (n, 3, 2))
    floor = np.concatenate([xy, np.zeros((n, 3, 1))], axis=2).astype(f32)   # many coplanar triangles
    tris = np.concatenate([tris_nasty(rs, 1500, 2.0), floor])
    # rays skimming the floor plane, plus feature-aimed and bounce-like rays
    o = np.concatenate([rs.uniform(-3, 3, (3000, 2)), 10.0 ** rs.uniform(-9, -2, (3000, 1)) * rs.choice([-1, 0, 1], (3000, 1))], axis=1)
    d = np.concatenate([rs.normal(size=(3000, 2)), 10.0 ** rs.uniform(-9, -2, (3000, 1)) * rs.choice([-1, 0, 1], (3000, 1))], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o1, d1 = rays_through_features(rs, tris, rs.uniform(-4, 4, (64, 3)), per_tri=2)
    o2, d2 = bounce_like_rays(rs, tris, 2000, 2.0, (0, 0, 0))
    fin = np.isfinite(o2).all(1) & np.isfinite(d2).all(1)
    st = run_clustered(fm, tris, np.concatenate([o.astype(f32), o1, o2[fin]]), np.concatenate([d.astype(f32), d1, d2[fin]]))
    assert st["bad"] == 0, st
    assert st["accepts"] > 1000, st


def test_clustered_path_rays_outside_the_records_range(fm):
    """Origins far outside the scene: the band, slab and sphere records do not apply, and every
    pair of such a ray must go to the edge filter."""
    rs = np.random.RandomState(74)
    tris = tris_random(rs, 500, 1.0, 0.2)
    far = rs.normal(size=(64, 3))
    far = far / np.linalg.norm(far, axis=1, keepdims=True) * rs.uniform(20, 2000, (64, 1))
    o, d = rays_through_features(rs, tris, far, per_tri=3)
    st = run_clustered(fm, tris, o, d)
    assert st["bad"] == 0, st
    assert st["wild"] == o.shape[0] and st["accepts"] > 300, st
    # and non-finite input never produces a wrong rejection
    o[0] = np.nan; d[1] = 0.0; o[2, 1] = np.inf; d[3, 2] = np.nan
    tris[0, 1] = tris[0, 0]; tris[1, 2, 0] = np.nan; tris[2, 1, 1] = np.inf
    st = run_clustered(fm, tris, o, d)
    assert st["bad"] == 0, st


def test_clustered_path_room_with_walls_and_small_objects(fm):
    """Two wall-sized triangles per side around small objects (the sword scene's shape): the walls
    are too large to cluster and stay in every band; the small triangles are clustered."""
    rs = np.random.RandomState(75)
    small = tris_random(rs, 1200, 1.0, 0.05)
    c = 3.0
    quads = []
    for axis in range(3):
        for sgn in (-1, 1):
            p = np.zeros((4, 3)); other = [a for a in range(3) if a != axis]
            p[:, axis] = sgn * c
            p[:, other[0]] = [-c, c, c, -c]; p[:, other[1]] = [-c, -c, c, c]
            quads += [p[[0, 1, 2]], p[[0, 2, 3]]]
    tris = np.concatenate([small, np.array(quads, dtype=f32)])
    o1, d1 = bounce_like_rays(rs, tris, 4000, 1.0, (0, 0, 0))
    o2, d2 = in_plane_rays(rs, tris, 4000, 2.0)
    st = run_clustered(fm, tris, np.concatenate([o1, o2]), np.concatenate([d1, d2]))
    assert st["bad"] == 0, st
    assert 12 <= st["unclustered"] < 40, st
    assert st["accepts"] > 3000, st


# -- cones of runs of 32 slots (the run skip of the shared-origin traversal) ---------------------
def run_cluster_cone(fm, tris, O, dirs, sign):
    c, r = bounding_sphere(tris)
    stats = (C.c_longlong * 6)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    O = np.ascontiguousarray(O, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    fm.fm_check_cluster_cone.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                         C.c_float, C.c_void_p]
    fm.fm_check_cluster_cone(tris.shape[0], tris.ctypes.data, dirs.shape[0], O.ctypes.data, dirs.ctypes.data,
                             c.ctypes.data, r, sign, stats)
    return dict(zip(["pairs", "run", "member", "bad", "everything", "nothing"], list(stats)))


def spatially_sorted(tris):
    """Runs of 32 are compact in space when the triangles are sorted along a curve."""
    c = tris.astype(np.float64).mean(1)
    q = ((c - c.min(0)) / (np.ptp(c, axis=0) + 1e-300) * 15).astype(int)
    return tris[np.lexsort((q[:, 2], q[:, 1], q[:, 0]))]


@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_run_cones_contain_their_members(fm, sc):
    rs = np.random.RandomState(81)
    tris = spatially_sorted(tris_random(rs, 2048, sc["extent"], sc["size"], sc["offset"]))
    for trial in range(3):
        O = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
        _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64), per_tri=2)
        # directions just outside and just inside member cones matter most: aim at the features
        # and jitter by 1e-7 .. 1e-2
        jit = dirs + rs.normal(size=dirs.shape).astype(f32) * (10.0 ** rs.uniform(-7, -2, (dirs.shape[0], 1))).astype(f32)
        for sign in (1.0, -1.0):
            st = run_cluster_cone(fm, tris, O, np.concatenate([dirs, jit]) * f32(sign), sign)
            assert st["bad"] == 0, st
            assert st["member"] > 1000, st
            if sc["name"] == "soup":
                assert st["run"] < 0.9 * st["pairs"] and st["everything"] < 0.1 * (tris.shape[0] // 32), st


def test_run_cones_with_degenerate_always_and_unsorted_members(fm):
    rs = np.random.RandomState(82)
    tris = tris_nasty(rs, 1024, 2.0)           # unsorted: runs span the scene, many cones fall back to "everything"
    O = f32([0.2, -0.1, 0.4])
    tris[5, 0] = O                             # a vertex at the origin: that member's cone is "everything"
    _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64), per_tri=3)
    fin = np.isfinite(dirs).all(1)
    st = run_cluster_cone(fm, tris, O, dirs[fin], 1.0)
    assert st["bad"] == 0 and st["everything"] >= 1, st


# ---------------------------------------------------------------------------------------------
# Direction cells: the per-cell triangle lists of the shared-origin traversal
# ---------------------------------------------------------------------------------------------
def run_grid(fm, tris, O, org, dirs, sign, tmax=None, bits=8):
    c, r = bounding_sphere(tris)
    stats = (C.c_longlong * 8)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    O = np.ascontiguousarray(O, dtype=f32)
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    tm = None if tmax is None else np.ascontiguousarray(tmax, dtype=f32)
    fm.fm_check_grid.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p]
    fm.fm_check_grid(tris.shape[0], tris.ctypes.data, org.shape[0], O.ctypes.data, org.ctypes.data,
                     dirs.ctypes.data, tm.ctypes.data if tm is not None else None, c.ctypes.data, r, sign, bits, stats)
    return dict(zip(["pairs", "listed", "both", "accepts", "bad", "cascade", "wild", "differ"], list(stats)))


def shadow_rays_towards(rs, tris, L, extent, per_tri=6, far=None):
    """Shadow rays as shadow_setup_kernel forms them, from surface points behind triangle features."""
    _, dirs = rays_through_features(rs, tris, L[None, :].astype(np.float64), per_tri=per_tri)
    dist = rs.uniform(0.05, 5.0, (dirs.shape[0], 1)) * extent if far is None else far
    p = (L[None, :].astype(np.float64) + dirs.astype(np.float64) * dist).astype(f32)
    dlt = (L[None, :] - p).astype(f32)
    inv = (f32(1.0) / np.sqrt((dlt * dlt).sum(1, dtype=f32))).astype(f32)
    ld = (dlt * inv[:, None]).astype(f32)
    org = (p + ld * f32(0.0000001)).astype(f32)
    pm = (p - ld * f32(0.0000001)).astype(f32)
    tmax = np.sqrt(((L[None, :] - pm) ** 2).sum(1, dtype=f32)).astype(f32)
    return org, ld, tmax


@pytest.mark.parametrize("bits", [4, 8, 9])
@pytest.mark.parametrize("sc", SCENES, ids=[s["name"] for s in SCENES])
def test_direction_cells_primary_rays(fm, sc, bits):
    rs = np.random.RandomState(91)
    tris = spatially_sorted(tris_random(rs, 1536, sc["extent"], sc["size"], sc["offset"]))
    for trial in range(2):
        O = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
        _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64))
        org = (O[None, :] + dirs * f32(0.001)).astype(f32)
        st = run_grid(fm, tris, O, org, dirs, 1.0, bits=bits)
        assert st["bad"] == 0 and st["differ"] == 0, st
        assert st["accepts"] > 1000 and st["wild"] == 0, st
        if sc["name"] == "soup" and bits >= 8:
            # the lists are short: far fewer pairs than the cone test alone lets through to the edge filter
            assert st["listed"] < 0.01 * st["pairs"], st


def test_direction_cells_rays_on_cell_borders_and_octahedron_seams(fm):
    """Directions on the grid lines of the octahedral map, on the fold |u|+|v| = 1, on the axes and
    on the seams of the lower hemisphere, a few ulps either side."""
    rs = np.random.RandomState(92)
    tris = spatially_sorted(tris_random(rs, 2048, 1.0, 0.08))
    O = f32([0.05, -0.02, 0.03])                     # inside the soup: triangles in every direction
    n = 6000
    G = 256.0
    u = (np.round(rs.uniform(-1, 1, n) * G / 2) * 2 / G)            # on grid lines
    v = rs.uniform(-1, 1, n)
    swap = rs.rand(n) < 0.5
    u, v = np.where(swap, v, u), np.where(swap, u, v)
    fold = rs.rand(n) < 0.25                                        # on the fold between the hemispheres
    v = np.where(fold, np.sign(v) * (1 - np.abs(u)), v)
    seam = rs.rand(n) < 0.15                                        # border of the square: lower-hemisphere seams
    u = np.where(seam, np.sign(u) + (u == 0), u)
    d = np.empty((n, 3))
    z = 1 - np.abs(u) - np.abs(v)
    lower = z < 0
    d[:, 0] = np.where(lower, (1 - np.abs(v)) * np.where(u >= 0, 1, -1), u)
    d[:, 1] = np.where(lower, (1 - np.abs(u)) * np.where(v >= 0, 1, -1), v)
    d[:, 2] = z
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d += rs.normal(size=d.shape) * 10.0 ** rs.uniform(-9, -6, (n, 1))
    axes = np.concatenate([np.eye(3), -np.eye(3)]).repeat(20, 0) + rs.normal(size=(120, 3)) * 1e-8
    d = np.concatenate([d, axes]).astype(f32)
    _, aimed = rays_through_features(rs, tris, O[None, :].astype(np.float64), per_tri=2)
    d = np.concatenate([d, aimed])
    org = (O[None, :] + d * f32(0.001)).astype(f32)
    for bits in (8, 9):
        st = run_grid(fm, tris, O, org, d, 1.0, bits=bits)
        assert st["bad"] == 0 and st["differ"] == 0, st
        assert st["accepts"] > 1000, st


def test_direction_cells_nasty_triangles_and_origins_on_planes(fm):
    rs = np.random.RandomState(93)
    tris = tris_nasty(rs, 2048, 2.0)
    n = 300
    xy = rs.uniform(-2, 2, (n, 3, 2))
    floor = np.concatenate([xy, np.zeros((n, 3, 1))], axis=2).astype(f32)
    O = f32([0.25, -0.5, 0.0])                       # in the floor's plane
    floor[0, 0] = O
    tris = np.concatenate([floor, tris])
    for origin in (O, f32([0.1, -0.2, 0.3]), f32([5.0, 1.0, -2.0])):
        _, dirs = rays_through_features(rs, tris, origin[None, :].astype(np.float64))
        fin = np.isfinite(dirs).all(1)
        dirs = dirs[fin]
        org = (origin[None, :] + dirs * f32(0.001)).astype(f32)
        st = run_grid(fm, tris, origin, org, dirs, 1.0)
        assert st["bad"] == 0 and st["differ"] == 0 and st["accepts"] > 1000, st


def test_direction_cells_edge_on_slivers_leave_the_lists(fm):
    """Triangles seen nearly edge-on have no cone; their edge planes still keep them out of the lists
    of cells that look elsewhere."""
    rs = np.random.RandomState(94)
    O = f32([0.0, 0.0, -3.0])
    n = 512
    c = rs.uniform(-1, 1, (n, 3))
    tris = np.empty((n, 3, 3))
    for i in range(n):
        toward = c[i] - O
        toward /= np.linalg.norm(toward)
        side = np.cross(toward, rs.normal(size=3))
        side /= np.linalg.norm(side)
        # spanned by the viewing direction and one sideways vector: the plane misses the origin by 1e-3..1e-2
        lift = np.cross(toward, side) * 10.0 ** rs.uniform(-3, -2)
        tris[i] = [c[i] + lift, c[i] + 0.08 * toward + lift, c[i] + 0.05 * side + 0.04 * toward + lift]
    tris = tris.astype(f32)
    _, dirs = rays_through_features(rs, tris, O[None, :].astype(np.float64))
    org = (O[None, :] + dirs * f32(0.001)).astype(f32)
    st = run_grid(fm, tris, O, org, dirs, 1.0)
    assert st["bad"] == 0 and st["differ"] == 0 and st["accepts"] > 200, st
    # every ray here is aimed at one of the slivers, whose wedges are long; without the plane test all pairs are listed
    assert st["listed"] < 0.3 * st["pairs"], st


@pytest.mark.parametrize("sc", SCENES[:3], ids=[s["name"] for s in SCENES[:3]])
def test_direction_cells_shadow_rays(fm, sc):
    rs = np.random.RandomState(95)
    tris = spatially_sorted(tris_random(rs, 1536, sc["extent"], sc["size"], sc["offset"]))
    L = (rs.uniform(-3, 3, 3) * sc["extent"] + np.asarray(sc["offset"])).astype(f32)
    org, ld, tmax = shadow_rays_towards(rs, tris, L, sc["extent"])
    st = run_grid(fm, tris, L, org, ld, -1.0, tmax)
    assert st["bad"] == 0 and st["differ"] == 0, st
    assert st["accepts"] > 300, st


def test_direction_cells_light_near_surfaces_and_wild_rays(fm):
    rs = np.random.RandomState(96)
    L = f32([0.3, 0.1, -0.2])
    tris = tris_random(rs, 1024, 1.0, 0.3)
    for i in range(200):   # planes through (or a hair off) the light
        a, b = rs.normal(size=3), rs.normal(size=3)
        off = rs.normal(size=3) * 10.0 ** rs.uniform(-9, -4)
        c = rs.uniform(-1, 1, 2)
        tris[i] = (L + off + np.array([c[0] * a + c[1] * b + 0.3 * rs.normal() * a,
                                       c[0] * a + c[1] * b + 0.3 * rs.normal() * b,
                                       c[0] * a + c[1] * b - 0.3 * rs.normal() * (a + b)])).astype(f32)
    org, ld, tmax = shadow_rays_towards(rs, tris, L, 1.0)
    st = run_grid(fm, tris, L, org, ld, -1.0, tmax)
    assert st["bad"] == 0 and st["differ"] == 0, st
    # rays that start far outside the scene bound meet every slot
    org, ld, _ = shadow_rays_towards(rs, tris, L, 1.0, per_tri=2, far=500.0)
    st = run_grid(fm, tris, L, org, ld, -1.0, np.full(ld.shape[0], 1e9, dtype=f32))
    assert st["bad"] == 0 and st["wild"] == ld.shape[0], st


def run_band_cells(fm, tris, org, dirs, bits=8, buckets=1):
    c, r = bounding_sphere(tris)
    stats = (C.c_longlong * 6)()
    tris = np.ascontiguousarray(tris, dtype=f32)
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    fm.fm_check_band_cells.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                       C.c_int, C.c_int, C.c_void_p]
    fm.fm_check_band_cells(tris.shape[0], tris.ctypes.data, org.shape[0], org.ctypes.data, dirs.ctypes.data,
                           c.ctypes.data, r, bits, buckets, stats)
    return dict(zip(["pairs", "listed", "passes", "bad", "no_bucket", "off_range"], list(stats)))


def plane_rays_per_triangle(rs, tris, n_per=4, tilt=(-9, -4)):
    """Rays that start in (or a hair off) a triangle's plane and run along it (or a hair out of it)."""
    t = tris.astype(np.float64)
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]
    nrm = np.cross(e1, e2)
    ok = np.linalg.norm(nrm, axis=1) > 1e-12
    t, e1, e2, nrm = t[ok], e1[ok], e2[ok], nrm[ok]
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    orgs, dirs = [], []
    for _ in range(n_per):
        a, b = rs.uniform(-3, 3, (2, t.shape[0], 1))
        o = t[:, 0] + a * e1 + b * e2 + nrm * rs.normal(size=(t.shape[0], 1)) * 10.0 ** rs.uniform(-9, -5, (t.shape[0], 1))
        a, b = rs.normal(size=(2, t.shape[0], 1))
        d = a * e1 + b * e2
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-300
        d += nrm * rs.normal(size=(t.shape[0], 1)) * 10.0 ** rs.uniform(*tilt, (t.shape[0], 1))
        orgs.append(o); dirs.append(d)
    return np.concatenate(orgs).astype(f32), np.concatenate(dirs).astype(f32)


@pytest.mark.parametrize("bits", [4, 8, 9])
def test_band_cells_hold_every_member_a_ray_is_inside(fm, bits):
    rs = np.random.RandomState(97)
    tris = np.concatenate([tris_random(rs, 1200, 1.0, 0.05), tris_nasty(rs, 300, 1.0)])
    org, dirs = plane_rays_per_triangle(rs, tris)
    keep = (np.abs(org).max(1) < 1.2) & np.isfinite(dirs).all(1)     # inside the records' range
    _, aimed = rays_through_features(rs, tris, org[:50].astype(np.float64), per_tri=1)
    org = np.concatenate([org[keep], org[keep][:aimed.shape[0]]])
    dirs = np.concatenate([dirs[keep], aimed[:org.shape[0] - keep.sum()]])
    # axis-parallel and seam directions too
    axes = np.concatenate([np.eye(3), -np.eye(3), [[1, 1, 0], [1, -1, 0], [0, 1, 1], [1, 0, -1]]]).astype(f32)
    org = np.concatenate([org, org[:axes.shape[0]]]); dirs = np.concatenate([dirs, axes])
    st = run_band_cells(fm, tris, org, dirs, bits=bits)
    assert st["bad"] == 0, st
    assert st["passes"] > 1000, st
    if bits >= 8:
        assert st["listed"] < 0.5 * st["pairs"], st   # the slivers among the nasty triangles are in every band


@pytest.mark.parametrize("buckets,offset", [(2, (0, 0, 0)), (8, (0, 0, 0)), (8, (40, -25, 3)), (16, (0.3, 0, 0))])
def test_band_cell_origin_buckets_hold_every_member_a_ray_is_inside(fm, buckets, offset):
    """The lists split by the origin of the ray: every member whose band and slab test a ray passes is on
    the list of the ray's direction cell AND origin bucket (rays that start on the mesh, as bounce rays do,
    in and a hair off the planes of triangles; a mesh far from the world's origin too)."""
    rs = np.random.RandomState(131 + buckets)
    tris = np.concatenate([tris_random(rs, 1200, 1.0, 0.05, offset), tris_nasty(rs, 300, 1.0) + np.asarray(offset, f32)])
    tris = tris.astype(f32)
    org, dirs = plane_rays_per_triangle(rs, tris, n_per=3)
    # origins pulled back onto (or next to) the triangles: inside the bounding sphere, where the buckets are
    t = np.repeat(tris[None], 3, 0).reshape(-1, 3, 3)[:org.shape[0]].astype(np.float64)
    w = rs.dirichlet([1, 1, 1], size=org.shape[0])[:, :, None]
    on = (t * w).sum(1)
    ok = np.isfinite(on).all(1) & np.isfinite(dirs).all(1)
    org2 = np.where(rs.uniform(size=(org.shape[0], 1)) < 0.7, on, 0.5 * (on + org)).astype(f32)
    org2, dirs2 = org2[ok], dirs[ok]
    # and rays between triangles: start on one, aimed along another's plane
    perm = rs.permutation(org2.shape[0])
    org3 = org2[perm]
    org_all = np.concatenate([org2, org3]); dirs_all = np.concatenate([dirs2, dirs2])
    st = run_band_cells(fm, tris, org_all, dirs_all, bits=8, buckets=buckets)
    assert st["bad"] == 0 and st["off_range"] == 0, st
    assert st["passes"] > 1000, st
    assert 0 < st["no_bucket"] < 0.3 * st["pairs"], st    # some rays start off the mesh: they read whole lists
    whole = run_band_cells(fm, tris, org_all, dirs_all, bits=8, buckets=1)
    assert st["listed"] <= whole["listed"], (st, whole)
    if buckets >= 8:
        # (not by much on this mesh: most of its listings are slivers whose slab is wide; on a mesh of
        # ordinary triangles the buckets shorten the lists several times over)
        assert st["listed"] < whole["listed"], (st, whole)
        tris = tris_random(rs, 2500, 1.0, 0.03, offset)
        on = (tris.astype(np.float64) * rs.dirichlet([1, 1, 1], size=tris.shape[0])[:, :, None]).sum(1).astype(f32)
        _, dirs = plane_rays_per_triangle(rs, tris, n_per=1)
        org = on[rs.permutation(on.shape[0])][:dirs.shape[0]]
        st = run_band_cells(fm, tris, org, dirs, bits=9, buckets=buckets)
        whole = run_band_cells(fm, tris, org, dirs, bits=9, buckets=1)
        assert st["bad"] == 0 and st["off_range"] == 0 and st["passes"] > 300, st
        assert st["listed"] < 0.4 * whole["listed"], (st, whole)


@pytest.mark.parametrize("G", [2, 8, 16])
def test_bucket_column_ranges_hold_every_bucket_that_meets_the_strip(fm, G):
    """band_bucket_kernel visits, per column of a cell's origin buckets, only the rows bucket_column_range()
    names: no bucket that meets the member's strip may lie outside (strips steep, flat, along a column,
    far off the box, with tiny and huge coefficients), and the ranges must not be the whole column either."""
    rs = np.random.RandomState(211 + G)
    n = 20000
    mag = 10.0 ** rs.uniform(-3, 4, (n, 1))
    ang = rs.uniform(0, 2 * np.pi, (n, 1))
    a, b = mag * np.cos(ang), mag * np.sin(ang)
    kind = rs.randint(0, 6, (n, 1))
    a = np.where(kind == 0, 0.0, a); b = np.where(kind == 1, 0.0, b)          # axis-parallel strips
    b = np.where(kind == 2, b * 1e-12, b)                                       # along a column, almost
    u0, v0 = rs.uniform(-3, 1, (2, n, 1)) * 10.0 ** rs.uniform(-1, 2, (2, n, 1))
    su, sv = 10.0 ** rs.uniform(-3, 1, (2, n, 1))
    # through a random point of the box (most), or somewhere else
    pu, pv = u0 + rs.uniform(-0.2, 1.2, (n, 1)) * su * G, v0 + rs.uniform(-0.2, 1.2, (n, 1)) * sv * G
    k = -(a * pu + b * pv) + np.where(kind == 3, mag * 100.0, 0.0)
    bound = 1.0 + 10.0 ** rs.uniform(-3, 2, (n, 1)) + (np.abs(a) * su + np.abs(b) * sv) * 0.501
    strips = np.ascontiguousarray(np.concatenate([k, a, b, bound, u0, v0, su, sv], 1), dtype=np.float64)
    st = np.zeros(3, dtype=np.int64)
    fm.fm_check_bucket_columns.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    fm.fm_check_bucket_columns(n, strips.ctypes.data, G, st.ctypes.data)
    meet, bad, ranged = (int(x) for x in st)
    assert bad == 0, (meet, bad, ranged)
    assert meet > n, (meet, ranged)
    if G >= 8:
        assert ranged < 0.9 * n * G * G, (meet, ranged)     # the ranges do cut the work (these strips are wide)


# -- the cells of a walk by rank (traverse_voxel_long_kernel) against the cells found step by step -----------
def run_walk_ranks(fm, gmin, cs, n, back, flat, org, dirs):
    org = np.ascontiguousarray(org, dtype=f32)
    dirs = np.ascontiguousarray(dirs, dtype=f32)
    gmin = np.ascontiguousarray(gmin, dtype=f32)
    n = np.ascontiguousarray(n, dtype=np.int32)
    st = np.zeros(8, dtype=np.int64)
    fm.fm_check_walk_ranks.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_void_p]
    fm.fm_check_walk_ranks(gmin.ctypes.data, cs, n.ctypes.data, back, flat, org.shape[0], org.ctypes.data,
                           dirs.ctypes.data, st.ctypes.data)
    return dict(walks=int(st[0]), states=int(st[1]), bad=int(st[2]), rounds=int(st[3]), outside=int(st[4]))


def walk_rays(rs, n, lo, hi):
    """Origins inside and around the box [lo, hi], directions of every kind: random, nearly and exactly
    axis-parallel, exactly diagonal (every step a tie between axes), through cell corners, tiny and huge."""
    span = hi - lo
    org = rs.uniform(lo - 0.3 * span, hi + 0.3 * span, (n, 3))
    d = rs.normal(size=(n, 3))
    kind = rs.randint(0, 8, n)
    ax = rs.randint(0, 3, n)
    for i in range(n):
        k = kind[i]
        if k == 1:      # exactly axis-parallel
            d[i] = 0.0; d[i, ax[i]] = rs.choice([-1.0, 1.0])
        elif k == 2:    # nearly axis-parallel: one or two components below / around the "flat" threshold
            d[i, (ax[i] + 1) % 3] *= 10.0 ** rs.uniform(-9, -3)
            d[i, (ax[i] + 2) % 3] *= 10.0 ** rs.uniform(-9, -3)
        elif k == 3:    # exact diagonals: equal parameters on two or three axes at every step
            d[i] = rs.choice([-1.0, 1.0], 3) * rs.choice([1.0, 1.0, 0.0], 3)
            if not d[i].any():
                d[i, 0] = 1.0
        elif k == 4:    # scaled: |d| far from 1
            d[i] *= 10.0 ** rs.uniform(-12, 12)
        elif k == 5:    # through (or a hair beside) a corner of the cell grid, from a point on the grid
            org[i] = lo + np.round((org[i] - lo) * 8) / 8
            d[i] = np.round(d[i] * 2) / 2
            if not d[i].any():
                d[i, 1] = -1.0
    return org.astype(f32), d.astype(f32)


def test_walk_by_rank_equals_walk_by_steps(fm):
    rs = np.random.RandomState(1234)
    total = rounds = walks = 0
    # grids of 64^3 cells of 1/32 (corners and diagonals exact in binary), of odd sizes, long and flat, offset far away
    for gmin, cs, n in [((-1.0, -1.0, -1.0), 1.0 / 32, (64, 64, 64)),
                        ((-1.07, -0.93, -1.21), 0.0371, (61, 57, 70)),
                        ((-3.0, -0.2, -0.1), 0.05, (130, 9, 3)),
                        ((999.0, -1001.0, 0.3), 0.0219, (128, 100, 77)),
                        ((-0.5, -0.5, -0.5), 1.0, (1, 1, 1))]:
        lo = np.array(gmin)
        hi = lo + cs * np.array(n)
        org, d = walk_rays(rs, 30000, lo, hi)
        for back, flat in ((0.08, 1e-7), (0.0, 0.0)):
            st = run_walk_ranks(fm, gmin, cs, n, back, flat, org, d)
            assert st["bad"] == 0, (gmin, cs, n, st)
            assert st["walks"] > 5000, st
            total += st["states"]
            rounds += st["rounds"]
            walks += st["walks"]
    assert total > 3_000_000
    assert rounds > 1.2 * walks   # walks of several rounds of 64 steps among them


# -- the seven lists of a cell in space ----------------------------------------------------------
@pytest.mark.parametrize("n_tris,res", [(600, 0), (3000, 12), (3000, 0), (40, 3)])
def test_cell_lists_hold_what_is_new_behind_each_face(fm, n_tris, res):
    """Every cell's list 6 is what the box / triangle test says, list f is list 6 without the face
    neighbour's list 6, no triangle twice, offsets of lists laid end to end (rt_voxel.h, build)."""
    rs = np.random.RandomState(77 + n_tris + res)
    c = rs.uniform(-1, 1, (n_tris, 1, 3))
    tris = (c + rs.uniform(-0.15, 0.15, (n_tris, 3, 3))).astype(np.float32)
    tris[::17] = (c[::17] + rs.uniform(-0.6, 0.6, (len(c[::17]), 3, 3))).astype(np.float32)  # some large ones
    tris = np.ascontiguousarray(tris.reshape(-1, 3))
    center = np.zeros(3, dtype=np.float32)
    stats = np.zeros(8, dtype=np.int64)
    fm.fm_check_cell_lists.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    fm.fm_check_cell_lists(n_tris, tris.ctypes.data, center.ctypes.data, C.c_float(2.2), res, stats.ctypes.data)
    cells, full, fresh, bad = (int(x) for x in stats[:4])
    assert bad == 0, stats
    assert cells > 0 and full > 0
    assert fresh < 6 * full  # the lists of new triangles are shorter than the whole lists
