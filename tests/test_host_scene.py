"""Camera, light rig, benchmark CSV line and light markers of the host library, against the
formulas of the reference's camera.cpp and main.cpp restated in numpy.

Parity unpinned: the reference has no fixtures for these; trigonometric results may differ from
a Windows build of the reference in the last bit (different libm)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from opengl_raytracer_amd import LIGHT_DTYPE, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
f32 = np.float32


@pytest.fixture(scope="module")
def host():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "opengl_raytracer_amd", "csrc", "host")])
    from opengl_raytracer_amd import host as h
    h.load()
    return h


@pytest.fixture(scope="module")
def classes():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "native"), "libhost_classes.so"])
    lib = C.CDLL(os.path.join(HERE, "native", "libhost_classes.so"))
    F = C.POINTER(C.c_float)
    lib.hc_camera_default.argtypes = [F]
    lib.hc_camera_pose.argtypes = [C.c_float] * 6 + [F]
    lib.hc_camera_input.argtypes = [C.c_double] + [C.c_int] * 4 + [C.c_double, C.c_double, C.c_int, F]
    lib.hc_mesh_group0.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), F]
    return lib


def cam_dump(fn, *args):
    out = (C.c_float * 46)()
    fn(*args, out)
    a = np.array(list(out), dtype=f32)
    return dict(pos=a[0:3], dir=a[3:6], right=a[6:9], up=a[9:12], h=a[12], v=a[13],
                view=a[14:30].reshape(4, 4).T, proj=a[30:46].reshape(4, 4).T)  # row-major math matrices


def ulps(a, b):
    a, b = np.asarray(a, f32), np.asarray(b, f32)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(np.spacing(np.maximum(np.abs(a), np.abs(b))), 1e-45)


# ------------------------------------------------------------------------------------- camera
@pytest.mark.parametrize("h,v", [(3.14, 0.0), (-82.44, 24.82), (0.3, -0.7), (100.0, 3.0)])
def test_camera_basis_matches_the_python_mirror(host, h, v):
    d, r, u = host.camera_basis(h, v)
    d2, u2, r2 = scenes.camera_basis(v, h)
    # cos/sin come from two different libms; everything else is the same float arithmetic
    assert np.abs(d - d2).max() < 3e-7 and np.abs(r - r2).max() < 3e-7 and np.abs(u - u2).max() < 5e-7
    assert r[1] == 0.0
    # up = cross(right, direction) exactly, in float (camera.cpp:65)
    x, y = r, d
    want = np.array([x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]], dtype=f32)
    assert np.array_equal(u, want)


def test_right_uses_3_14_not_pi(host):
    """camera.cpp:58-62 subtracts 3.14f/2, so right is not exactly perpendicular to direction."""
    d, r, _ = host.camera_basis(0.0, 0.0)
    assert abs(float(np.dot(d.astype(np.float64), r.astype(np.float64))) - math.cos(-1.57)) < 1e-6
    assert abs(float(np.dot(d, r))) > 5e-4


def test_matrices(host):
    p = host.perspective(90.0, 4.0 / 3.0, 0.1, 150.0).reshape(4, 4).T
    t = math.tan(45.0)  # the reference passes 90 to a function that takes radians (camera.cpp:84-87)
    want = np.zeros((4, 4))
    want[0, 0], want[1, 1] = 1 / (4 / 3 * t), 1 / t
    want[2, 2], want[2, 3], want[3, 2] = -(150.1) / 149.9, -2 * 150 * 0.1 / 149.9, -1
    assert np.allclose(p, want, rtol=2e-6, atol=0)
    eye, center, up = np.array([1.0, 2.0, 3.0]), np.array([0.5, 1.0, -2.0]), np.array([0.1, 1.0, 0.0])
    m = host.look_at(eye, center, up).reshape(4, 4).T
    f = (center - eye) / np.linalg.norm(center - eye)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    want = np.eye(4)
    want[0, :3], want[1, :3], want[2, :3] = s, u, -f
    want[:3, 3] = [-s @ eye, -u @ eye, f @ eye]
    assert np.allclose(m, want, atol=2e-6)
    assert np.allclose(m @ np.append(eye, 1), [0, 0, 0, 1], atol=2e-6)


def test_camera_class_defaults(classes, host):
    """Camera::Camera() (camera.cpp:6-18): origin, horizontal 3.14, vertical 0, FoV 90, 4:3."""
    c = cam_dump(classes.hc_camera_default)
    assert np.array_equal(c["pos"], [0, 0, 0]) and c["h"] == f32(3.14) and c["v"] == 0
    d, r, u = host.camera_basis(3.14, 0.0)
    assert np.array_equal(c["dir"], d) and np.array_equal(c["right"], r) and np.array_equal(c["up"], u)
    assert np.array_equal(c["proj"].T.reshape(-1), host.perspective(90.0, f32(4.0) / f32(3.0), 0.1, 150.0))
    assert np.array_equal(c["view"].T.reshape(-1), host.look_at(c["pos"], c["pos"] + c["dir"], c["up"]))


def test_camera_class_benchmark_pose(classes):
    """The pose RunRaytracer sets (main.cpp:402-406) equals the Python mirror's default camera."""
    c = cam_dump(classes.hc_camera_pose, 2.77904, 0.594951, -1.95046, 24.82, -82.44, 1.0)
    pos, d, u, r = scenes.default_camera()
    assert np.array_equal(c["pos"], pos)
    assert np.abs(c["dir"] - d).max() < 3e-7 and np.abs(c["up"] - u).max() < 5e-7 and np.abs(c["right"] - r).max() < 3e-7
    assert c["proj"][0, 0] == c["proj"][1, 1]  # aspect 1


def test_camera_class_input(classes):
    """Keys move at speed 3 per second along direction / right; the mouse turns at 0.005 rad per
    pixel from the window centre (camera.cpp:12-13, 47-48, 68-82)."""
    still = cam_dump(classes.hc_camera_input, 0.5, 0, 0, 0, 0, 0.0, 0.0, 0)
    assert np.array_equal(still["pos"], [0, 0, 0])
    fwd = cam_dump(classes.hc_camera_input, 0.5, 1, 0, 0, 0, 0.0, 0.0, 0)
    assert np.allclose(fwd["pos"], still["dir"] * 1.5, atol=1e-6)
    back_left = cam_dump(classes.hc_camera_input, 0.25, 0, 1, 0, 1, 0.0, 0.0, 0)
    assert np.allclose(back_left["pos"], -(still["dir"] + still["right"]) * 0.75, atol=1e-6)
    both = cam_dump(classes.hc_camera_input, 0.25, 1, 1, 1, 1, 0.0, 0.0, 0)
    assert np.allclose(both["pos"], [0, 0, 0], atol=1e-6)
    # the first Update already sees the cursor at the centre, the second 40 px left / 20 px up
    turned = cam_dump(classes.hc_camera_input, 0.0, 0, 0, 0, 0, -40.0, -20.0, 1)
    assert abs(turned["h"] - (3.14 + 0.005 * 40)) < 1e-6 and abs(turned["v"] - 0.005 * 20) < 1e-6
    frozen = cam_dump(classes.hc_camera_input, 0.0, 0, 0, 0, 0, -40.0, -20.0, 0)
    assert frozen["h"] == f32(3.14) and frozen["v"] == 0


def test_mesh_class(classes):
    n, size, first = C.c_int(), C.c_int(), (C.c_float * 8)()
    d = os.path.join(ROOT, "assets", "sword") + os.sep
    groups = classes.hc_mesh_group0(d.encode(), b"sword_fixed.obj", C.byref(n), C.byref(size), first)
    assert groups == 1 and n.value == 495 and size.value == 495 * 32
    assert np.isfinite(np.array(list(first))).all()
    assert classes.hc_mesh_group0(d.encode(), b"missing.obj", C.byref(n), C.byref(size), first) == -1


# ------------------------------------------------------------------------------------- lights
def test_default_lights(host):
    got = host.default_lights()
    want = scenes.default_lights()
    assert got.dtype == LIGHT_DTYPE and got.shape == (10,)
    assert np.array_equal(got["color"], want["color"])          # MSVC rand(), exact
    assert np.abs(got["pos"] - want["pos"]).max() < 5e-7        # cosf/sinf of two libms
    assert (got["pad0"] == -1).all() and (got["pad1"] == -1).all()
    assert np.array_equal(got["pos"][0], [2, 0, -2]) and (got["pos"][:, 2] == -2).all()
    assert np.allclose(np.hypot(got["pos"][:, 0], got["pos"][:, 1]), 2, atol=1e-6)
    # first colour: rand() = 41, 18467, 6334 over 32767 (the C runtime's documented first draws)
    assert np.array_equal(got["color"][0], (np.array([41, 18467, 6334], dtype=np.float64) / 32767.0).astype(f32))


def test_light_rotation(host):
    lights = host.default_lights()
    before = lights.copy()
    off = host.light_angle_offset(1.0)
    assert abs(off - math.sin(1.0) * 2 * math.pi / 10) < 1e-7 and host.light_angle_offset(0.0) == 0.0
    host.set_light_positions(lights, 4, off)
    ang = np.arctan2(lights["pos"][:4, 1].astype(np.float64), lights["pos"][:4, 0])
    assert np.allclose(ang, off + np.arange(4) * 2 * math.pi / 10, atol=1e-6)
    assert np.array_equal(lights["pos"][4:], before["pos"][4:]) and np.array_equal(lights["color"], before["color"])
    want = scenes.default_lights(angle_offset=off)
    assert np.abs(lights["pos"][:4] - want["pos"][:4]).max() < 5e-7


# ------------------------------------------------------------------------------------- CSV
def test_csv_line_format(host, tmp_path):
    """WriteBenchmarkResultsToCSVFile (main.cpp:209-219): nine integers, comma separated, appended."""
    p = str(tmp_path / "resolution_results.csv")
    host.csv_append(p, 32, 800, 800, 2, 2, 165, 0, 12, 7)
    host.csv_append(p, 64, 3840, 2160, 6, 3, 65536, 1.9, 150.7, 61.2)  # times truncate like the int columns
    assert open(p).read() == "32,800,800,2,2,165,0,12,7\n64,3840,2160,6,3,65536,1,150,61\n"
    with pytest.raises(host.HostError):
        host.csv_append(str(tmp_path / "no" / "dir.csv"), 1, 1, 1, 1, 1, 1, 1, 1, 1)


# ------------------------------------------------------------------------------------- markers
def test_light_markers(host):
    ident = np.eye(4, dtype=f32).reshape(-1)
    lights = np.zeros(3, dtype=LIGHT_DTYPE)
    lights["pos"] = [(0, 0, 0), (1.0, 1.0, 0), (0.5, 0.0, 2.0)]  # centre, top-right corner, beyond the far plane
    frame = np.zeros((100, 200, 4), dtype=np.uint8)
    host.draw_light_markers(frame, ident, ident, lights, 3, 30.0)
    yellow = (frame == np.array([255, 255, 0, 255], dtype=np.uint8)).all(-1)
    assert yellow[35:65, 85:115].all() and yellow[35:65, 85:115].sum() == 900
    assert yellow[85:, 185:].all()                      # the corner marker is clipped to the frame
    assert yellow.sum() == 900 + 15 * 15                # and the third light is outside the clip volume
    assert not frame[~yellow].any()
    frame[:] = 7
    host.draw_light_markers(frame, ident, ident, lights, 0, 30.0)
    assert (frame == 7).all()


def test_light_markers_through_the_camera(host, classes):
    """With the benchmark pose the forward pass projects through perspective(90 rad) * lookAt."""
    c = cam_dump(classes.hc_camera_pose, 0.0, 0.0, -5.0, 0.0, 0.0, 1.0)   # looking down +z
    lights = host.default_lights()
    frame = np.zeros((400, 400, 4), dtype=np.uint8)
    host.draw_light_markers(frame, c["view"].T.reshape(-1).copy(), c["proj"].T.reshape(-1).copy(), lights, 10, 30.0)
    yellow = (frame[..., 0] == 255)
    # every light is in front of the camera at z = -2 (distance 3) on a circle of radius 2
    t = math.tan(45.0)
    for l in lights["pos"]:
        ndc = np.array([l[0], l[1]]) / (3.0 * t)
        # the view matrix mirrors x for a camera looking down +z (right = (sin(h - 1.57), 0, ...))
        px = (np.array([-ndc[0], ndc[1]]) * 0.5 + 0.5) * 400
        assert yellow[int(px[1]), int(px[0])], (l, px)
