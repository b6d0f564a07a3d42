"""rt_read_rays_static -- the entry point that allocates nothing -- against rt_read_rays, through the
C ABI: the same 80-byte records, on two calls in a row (the second from a smaller context, so the
static staging array still holds the first call's records beyond the second call's), then the size
limit.  Collected last: it must pass on its own and after every other test."""
import numpy as np
import pytest

from common import setup, small_scene

pytestmark = pytest.mark.gpu


def test_read_rays_static_equals_read_rays(rt):
    sc = small_scene("mixed")
    with rt.RayTracer(512, 400) as big, rt.RayTracer(96, 64) as small:
        for g in (big, small):
            setup(g, sc, 2)
            g.render(2)
        for g in (big, small, big):          # large, then smaller, then large again
            want = g.read_rays()
            got = g.read_rays_static()
            assert got.tobytes() == want.tobytes()       # every byte of the 80-byte records, padding included
            assert (want["primitiveID"] >= 0).any()
        # a share of the large frame: fewer rays through the same context
        big.set_row_tiles(8, 1, 5)
        big.render(2)
        assert big.read_rays_static().tobytes() == big.read_rays().tobytes()


def test_read_rays_static_rejects_frames_beyond_the_array(rt):
    sc = small_scene("mixed")
    with rt.RayTracer(1200, 800) as g:       # 960 000 rays > 838 860
        setup(g, sc, 1)
        g.render(1)
        with pytest.raises(rt.RtError) as e:
            g.read_rays_static()
        assert e.value.code == -1 and "static staging array" in str(e.value)
        g.set_row_tiles(16, 0, 2)            # half the rows fit
        g.render(1)
        assert g.read_rays_static().tobytes() == g.read_rays().tobytes()
