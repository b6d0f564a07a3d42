"""The C-ABI library builds, loads and exports every symbol include/rt.h declares.
No compute call is made here (there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="rt.h", prefix="rt_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_three_stage_calls():
    syms = declared_symbols()
    for s in ("rt_raygen", "rt_intersect", "rt_color", "rt_create", "rt_destroy", "rt_set_mesh"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from opengl_raytracer_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_native.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"librt_hip.so does not export {s}"
    # and the binding declares a prototype for each of them
    assert sorted(_native.PROTOTYPES) == declared_symbols()
    assert lib.rt_version is not None
    lib.rt_version.restype = C.c_char_p
    assert b"gfx950" in lib.rt_version()


def test_host_library_exports_every_declared_symbol():
    """librt_host.so (OBJ/MTL, images, camera, lights, CSV, markers) against include/rt_host.h."""
    import subprocess
    from opengl_raytracer_amd import host
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "opengl_raytracer_amd", "csrc", "host")])
    lib = C.CDLL(host.LIB_PATH)
    syms = declared_symbols("rt_host.h", "rth_")
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"librt_host.so does not export {s}"
    assert sorted(host.PROTOTYPES) == syms
    assert C.sizeof(host.Material) == 128 + 4 + 36 + 256 + 4


def test_record_sizes_match_the_reference_layouts():
    from opengl_raytracer_amd import LIGHT_DTYPE, RAY_DTYPE, VERTEX_DTYPE
    # Ray.h:9-24 / definitions.glsl:1-10
    assert RAY_DTYPE.itemsize == 80
    assert [RAY_DTYPE.fields[k][1] for k in ("origin", "dir", "color", "t", "primitiveID", "n")] == \
        [0, 16, 32, 44, 48, 64]
    # main.cpp:150-167
    assert VERTEX_DTYPE.itemsize == 32 and VERTEX_DTYPE.fields["texCoord"][1] == 16
    # main.cpp:29-35
    assert LIGHT_DTYPE.itemsize == 32 and LIGHT_DTYPE.fields["color"][1] == 16


def test_create_fails_loudly_without_a_gpu():
    """On a box without a HIP device the product must raise, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import opengl_raytracer_amd as pkg
    with pytest.raises(pkg.RtError):
        pkg.RayTracer(8, 8)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "opengl_raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "rt_oracle" not in text and "rto_" not in text, f
