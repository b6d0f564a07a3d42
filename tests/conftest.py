import os
import sys

import pytest

# The library sends the bounce rays of meshes under 1024 triangles through the per-pair filter.  The
# small scenes of these tests are there to exercise the group sweeps, so the threshold is lifted;
# test_small_mesh_default_threshold covers the default.
os.environ.setdefault("RT_CLUSTER_MIN_TRIS", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
