#!/usr/bin/env python3
"""Stores what Pillow decodes from the reference's two texture files, as the expected output for
tests/test_host_image.py::test_reference_data_files.  Run where /root/reference exists:

    python tests/golden/images/make_image_golden.py

bth.tga (786 bytes) is copied next to its expected array; sword.png is the copy under assets/."""
import os
import shutil
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/OpenGL Raytracer"

shutil.copyfile(os.path.join(REF, "bth_logo_obj_tga", "bth.tga"), os.path.join(HERE, "bth.tga"))
np.save(os.path.join(HERE, "bth_tga_rgba.npy"),
        np.array(Image.open(os.path.join(HERE, "bth.tga")).convert("RGBA")))

a = np.array(Image.open(os.path.join(REF, "sword", "sword.png")).convert("RGBA"))[::-1].copy()
np.savez(os.path.join(HERE, "sword_png_digest.npz"), shape=np.array(a.shape),
         crc32_bottom_up=np.uint32(zlib.crc32(a.tobytes())), samples_bottom_up=a[::64, ::64])
print("bth.tga", Image.open(os.path.join(HERE, "bth.tga")).size, "sword.png", a.shape)
