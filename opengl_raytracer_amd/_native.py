"""ctypes binding of librt_hip.so (the C ABI declared in include/rt.h).

There is no fallback: if the HIP library has not been built, or no gfx950 device
is present when a context is created, this raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIB selects another build of the same library (A/B timing of kernel variants)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")

# Record layouts of include/rt.h ("OpenGL Raytracer/Ray.h":9-24, main.cpp:29-35, 148-167)
RAY_DTYPE = np.dtype([
    ("origin", "<f4", 3), ("pad0", "<f4"),
    ("dir", "<f4", 3), ("pad1", "<f4"),
    ("color", "<f4", 3), ("t", "<f4"),
    ("primitiveID", "<i4"), ("pad2", "<f4", 3),
    ("n", "<f4", 3), ("pad5", "<f4"),
])
VERTEX_DTYPE = np.dtype([
    ("pos", "<f4", 3), ("pad0", "<f4"),
    ("texCoord", "<f4", 2), ("pad1", "<f4", 2),
])
LIGHT_DTYPE = np.dtype([
    ("pos", "<f4", 3), ("pad0", "<f4"),
    ("color", "<f4", 3), ("pad1", "<f4"),
])
assert RAY_DTYPE.itemsize == 80 and VERTEX_DTYPE.itemsize == 32 and LIGHT_DTYPE.itemsize == 32

KERNEL_RECORD_DTYPE = np.dtype([
    ("stage", "<i4"), ("form", "<i4"), ("rays_per_lane", "<i4"), ("exact", "<i4"),
    ("rays", "<u8"), ("pairs", "<u8"), ("ms", "<f4"), ("reserved", "<f4"),
])
assert KERNEL_RECORD_DTYPE.itemsize == 40

RT_MAX_LIGHTS = 10
RT_TRACE_FILTERED = 0
RT_TRACE_EXACT = 1
RT_TRACE_CLUSTERED = 2
# rt_option (include/rt.h)
OPTIONS = {"sort_min_rays": 0, "cluster_min_tris": 1, "grid_bits": 2, "sweeps_in_turn": 3, "band_list_max": 4,
           "voxel_res": 5, "walk_budget": 6, "fail_next_stage": 7, "moving_origin_levels": 8,
           "kernel_timing": 9, "band_buckets": 10, "exact_rays_per_lane": 11}

# every symbol include/rt.h declares: name -> (restype, argtypes)
_P = C.c_void_p
PROTOTYPES = {
    "rt_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int, C.c_int]),
    "rt_destroy": (None, [_P]),
    "rt_set_mesh": (C.c_int, [_P, _P, C.c_int]),
    "rt_set_texture_rgba8": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "rt_set_lights": (C.c_int, [_P, _P, C.c_int]),
    "rt_set_camera": (C.c_int, [_P, _P, _P, _P, _P]),
    "rt_set_row_tiles": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "rt_local_rows": (C.c_int, [_P]),
    "rt_num_rays": (C.c_int, [_P]),
    "rt_get_row_map": (C.c_int, [_P, _P]),
    "rt_set_trace_mode": (C.c_int, [_P, C.c_int]),
    "rt_set_option": (C.c_int, [_P, C.c_int, C.c_longlong]),
    "rt_raygen": (C.c_int, [_P]),
    "rt_intersect": (C.c_int, [_P]),
    "rt_color": (C.c_int, [_P]),
    "rt_render": (C.c_int, [_P, C.c_int]),
    "rt_sync": (C.c_int, [_P]),
    "rt_read_rays": (C.c_int, [_P, _P]),
    "rt_read_rays_static": (C.c_int, [_P, _P]),
    "rt_write_rays": (C.c_int, [_P, _P]),
    "rt_read_pixels_rgba8": (C.c_int, [_P, _P]),
    "rt_read_depth": (C.c_int, [_P, _P]),
    "rt_stage_ms": (C.c_int, [_P, _P]),
    "rt_get_counters": (C.c_int, [_P, _P]),
    "rt_kernel_log": (C.c_int, [_P, _P, C.c_int, _P]),
    "rt_bind_stream": (C.c_int, [_P, _P]),
    "rt_unbind_stream": (C.c_int, [_P]),
    "rt_bind_framebuffer": (C.c_int, [_P, _P]),
    "rt_last_error": (C.c_char_p, [_P]),
    "rt_version": (C.c_char_p, []),
}

_lib = None


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rt error {code}: {message}")
        self.code = code


def _elf_dynamic(path):
    """(DT_SONAME or None, [DT_NEEDED names]) of a 64-bit little-endian ELF file, read from its dynamic section."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:6] != b"\x7fELF\x02\x01":
        raise ValueError(f"{path}: not a 64-bit little-endian ELF file")
    phoff, = struct.unpack_from("<Q", data, 0x20)
    phentsize, phnum = struct.unpack_from("<HH", data, 0x36)
    loads, dyn = [], None
    for i in range(phnum):
        ptype, _flags, off, vaddr, _paddr, filesz, _memsz, _align = struct.unpack_from("<IIQQQQQQ", data, phoff + i * phentsize)
        if ptype == 1:
            loads.append((vaddr, off, filesz))
        elif ptype == 2:
            dyn = (off, filesz)
    if dyn is None:
        return None, []
    entries = [struct.unpack_from("<qQ", data, dyn[0] + k) for k in range(0, dyn[1], 16)]
    strtab = next((v for t, v in entries if t == 5), None)
    if strtab is None:
        return None, []
    base = next((off + strtab - vaddr for vaddr, off, filesz in loads if vaddr <= strtab < vaddr + filesz), None)
    if base is None:
        return None, []

    def name(at):
        end = data.index(b"\0", base + at)
        return data[base + at:end].decode()

    soname = next((name(v) for t, v in entries if t == 14), None)
    return soname, [name(v) for t, v in entries if t == 1]


def _share_hip_runtime_with_torch():
    """A process can use the GPU through ONE copy of the HIP runtime only: a second copy finds no
    device.  PyTorch's ROCm wheels carry their own (torch/lib/libamdhip64.so), and a later `import torch`
    in the same process loads it.  So when torch is installed but not imported yet, its copy is loaded
    here first -- without importing torch -- provided its soname is the one librt_hip.so asks for
    (DT_NEEDED), so that the library's dependency resolves to it; then torch finds the runtime it expects.
    If the sonames differ the two would not share a runtime and this raises, rather than let one of them
    find "no HIP GPUs" later.  RT_NO_TORCH_HIP_PRELOAD=1 skips all of this (for a process that never
    imports torch and wants the system's runtime)."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("RT_NO_TORCH_HIP_PRELOAD") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return
    wanted = [n for n in _elf_dynamic(LIB_PATH)[1] if n.startswith("libamdhip64.so")]
    soname = _elf_dynamic(path)[0]
    if wanted and soname not in wanted:
        raise ImportError(
            f"{LIB_PATH} was built against {wanted[0]} but the installed torch carries {soname} ({path}): one process "
            "cannot use both.  Rebuild the library against torch's runtime, or set RT_NO_TORCH_HIP_PRELOAD=1 and "
            "do not import torch in this process")
    C.CDLL(path, mode=C.RTLD_GLOBAL)


def load():
    """Loads librt_hip.so and sets the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C opengl_raytracer_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
