import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The library sends the bounce rays of meshes under 1024 triangles through the per-pair filter.  The
# small scenes of these tests are there to exercise the group sweeps, so every context of the suite
# starts with that threshold at 0 (rt_set_option); test_small_mesh_default_threshold covers the default.
from opengl_raytracer_amd import raytracer as _raytracer  # noqa: E402

_raytracer.DEFAULT_OPTIONS["cluster_min_tris"] = 0


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import rt_oracle
    rt_oracle.load()
    return rt_oracle


@pytest.fixture(scope="session")
def rt():
    """The product package; GPU tests only."""
    import opengl_raytracer_amd as pkg
    return pkg
