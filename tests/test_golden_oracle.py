"""The oracle reproduces the committed golden vectors bit for bit (guards the oracle
and the seeded scene builders against drift)."""
import glob
import os

import numpy as np
import pytest

from common import rays_equal_bits, describe_mismatch
from oracle.rt_oracle import Oracle

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_case(path):
    z = np.load(path)
    return {k: z[k] for k in z.files}


def drive(obj, case, check):
    """Runs the case's stage sequence on obj, calling check(tag, got, want) at every dump."""
    w, h, passes, nl, tr, tf, ts = [int(x) for x in case["meta"]]
    obj.set_mesh(case["mesh"])
    obj.set_texture(case["tex"])
    obj.set_lights(case["lights"], nl)
    obj.set_camera(*case["cam"])
    if tr:
        obj.set_row_tiles(tr, tf, ts)
    obj.raygen()
    check("rays_g", obj.read_rays(), case["rays_g"])
    for p in range(passes):
        obj.intersect()
        check(f"rays_i{p}", obj.read_rays(), case[f"rays_i{p}"])
        obj.color()
        check(f"rays_c{p}", obj.read_rays(), case[f"rays_c{p}"])
        check(f"pix_c{p}", obj.read_pixels(), case[f"pix_c{p}"])
    check("depth", obj.read_depth(), case["depth"])


def exact_check(tag, got, want):
    if tag.startswith("rays"):
        bad = rays_equal_bits(got, want)
        assert not bad.any(), tag + ": " + describe_mismatch(got, want, bad)
    elif tag == "depth":
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), tag
    else:
        assert np.array_equal(got, want), tag


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_golden(path):
    case = load_case(path)
    w, h = int(case["meta"][0]), int(case["meta"][1])
    drive(Oracle(w, h), case, exact_check)


def test_golden_cases_are_not_trivial():
    assert len(GOLDEN) >= 4
    for path in GOLDEN:
        case = load_case(path)
        ids = case["rays_i0"]["primitiveID"]
        assert (ids >= 0).mean() > 0.05, path          # a good share of rays hit something
        assert (ids == -2).mean() > 0.05, path         # and a good share miss
        pix = case["pix_c0"].reshape(-1, 4)
        assert len(np.unique(pix, axis=0)) > 8, path   # shaded, not flat
