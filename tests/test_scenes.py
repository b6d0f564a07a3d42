"""Host-side inputs: camera basis, default lights, scene builders (no GPU)."""
import numpy as np

from opengl_raytracer_amd import scenes


def test_default_camera_matches_camera_cpp():
    # camera.cpp:50-65 evaluated in float32 for vertical 24.82, horizontal -82.44 (main.cpp:402-405)
    pos, d, up, right = scenes.default_camera()
    assert np.allclose(pos, [2.77904, 0.594951, -1.95046])
    assert np.allclose(d, [-0.6545337, -0.30766836, 0.6905982], atol=2e-6)
    assert np.allclose(right, [-0.7263521, 0.0, -0.68732285], atol=2e-6)
    assert np.allclose(up, [-0.21146749, 0.9514934, 0.22347556], atol=2e-6)
    # right uses 3.14f/2, so it is not exactly orthogonal to the direction
    assert abs(float(d @ right)) > 1e-4


def test_default_lights_match_main_cpp():
    l = scenes.default_lights()
    # MSVC rand() seed 1: 41, 18467, 6334 -> light 0 colour (main.cpp:271-274, 383)
    g = scenes.msvc_rand(1)
    assert [next(g) for _ in range(6)] == [41, 18467, 6334, 26500, 19169, 15724]
    assert np.allclose(l[0]["color"], [41 / 32767, 18467 / 32767, 6334 / 32767])
    # radius-2 circle at z = -2, angle step 2*pi/10 (main.cpp:276-285)
    assert np.allclose(l[0]["pos"], [2, 0, -2])
    assert np.allclose(l[1]["pos"], [1.618034, 1.175571, -2], atol=1e-5)
    assert np.allclose(l[2]["pos"], [0.618034, 1.902113, -2], atol=1e-5)
    assert (l["pad0"] == -1).all() and (l["pad1"] == -1).all()


def test_scene_builders_are_deterministic_and_sized():
    a = scenes.triangle_soup(1000, seed=5)
    b = scenes.triangle_soup(1000, seed=5)
    assert a.tobytes() == b.tobytes() and a.shape[0] == 3000
    assert scenes.blob().shape[0] == 3 * 28576
    q = scenes.quad_grid(3, 2)
    assert q.shape[0] == 3 * 2 * 2 * 3
    assert scenes.noise_texture(8, 4).shape == (4, 8, 4)
