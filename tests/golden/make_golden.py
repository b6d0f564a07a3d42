"""Generates the golden vectors in this directory with the CPU oracle.

    python tests/golden/make_golden.py

Each .npz holds the inputs (mesh, texture, lights, camera, frame size, passes, lights
used, row tiling) and the expected outputs: the 80-byte ray records after every stage
call, the RGBA8 frame after every color call and the depth plane at the end.
The reference ships no vectors of its own; these pin the product against the oracle,
which is in turn pinned by tests/test_oracle_kat.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

from common import setup, small_scene  # noqa: E402
from oracle.rt_oracle import Oracle  # noqa: E402

CASES = [
    # name, scene kind, width, height, passes, lights, (tile_rows, first, stride)
    ("mixed_p2_l2", "mixed", 48, 48, 2, 2, None),
    ("soup_p3_l1", "soup", 48, 40, 3, 1, None),
    ("blob_p1_l1", "blob", 56, 32, 1, 1, None),
    ("mixed_p2_l1_tiles", "mixed", 40, 48, 2, 1, (4, 1, 3)),
]


def run_case(kind, w, h, passes, nl, tiles):
    sc = small_scene(kind)
    o = Oracle(w, h)
    setup(o, sc, nl)
    if tiles:
        o.set_row_tiles(*tiles)
    out = {}
    o.raygen()
    out["rays_g"] = o.read_rays()
    for p in range(passes):
        o.intersect()
        out[f"rays_i{p}"] = o.read_rays()
        o.color()
        out[f"rays_c{p}"] = o.read_rays()
        out[f"pix_c{p}"] = o.read_pixels()
    out["depth"] = o.read_depth()
    cam = np.stack([np.asarray(v, dtype=np.float32) for v in sc["cam"]])
    out.update(mesh=sc["mesh"], tex=sc["tex"], lights=sc["lights"], cam=cam,
               meta=np.array([w, h, passes, nl] + (list(tiles) if tiles else [0, 0, 0]), dtype=np.int32))
    return out


if __name__ == "__main__":
    for name, kind, w, h, passes, nl, tiles in CASES:
        out = run_case(kind, w, h, passes, nl, tiles)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, os.path.getsize(path) // 1024, "KiB")
