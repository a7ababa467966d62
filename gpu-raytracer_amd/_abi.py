"""ctypes binding of the okrt C ABI (include/okrt.h).

`DeviceLibrary` binds the product, libokrt.so (HIP, gfx950). It fails loudly when the shared
library is missing or does not export every symbol the header declares: there is no CPU fallback
in the product. `bind(lib, prefix)` applies the same signature table to any library that exports
these entry points under another prefix (the test suite uses that for its CPU checker).
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_DIR = os.path.dirname(PKG_DIR)

OKRT_ABI_VERSION = 4

# okrt_table slots = the reference's t-registers (ShaderResourceRegisters.h:9-21)
TABLE_TRI_POS, TABLE_TRI_INFO, TABLE_BVH = 0, 1, 2
TABLE_OCT, TABLE_SPHERE, TABLE_MESH = 5, 6, 7
TABLE_DIR_LIGHT, TABLE_POINT_LIGHT, TABLE_SPOT_LIGHT = 8, 9, 10
TABLE_STRIDE = {0: 36, 1: 132, 2: 36, 5: 48, 6: 84, 7: 232, 8: 32, 9: 32, 10: 48}

OPT_TRAVERSAL, OPT_MAX_BOUNCES, OPT_COUNTERS, OPT_TIMING, OPT_BUILDER, OPT_DEFER, OPT_QUEUE_MEMORY_MB, OPT_DEBUG_FAIL_ALLOC, OPT_PIPELINE = 0, 1, 2, 3, 4, 5, 6, 7, 8
OPT_LAST_SEGMENT_SHORTCUT, OPT_LAST_SEGMENT_ANY_HIT = 9, 10
BUILDER_HOST, BUILDER_DEVICE = 0, 1
TRAVERSAL_REFERENCE, TRAVERSAL_FAST, TRAVERSAL_AUTO = 0, 1, 2
BUILD_FLAG_SCALED_INSTANCES, BUILD_FLAG_SHRUNK_INSTANCES, BUILD_FLAG_DARK_MATERIALS, BUILD_FLAG_CAMERA_PER_RAY = 1, 2, 4, 8


class RenderData(C.Structure):
    """b0 cbuffer, 240 bytes (RaytracerCS.hlsl:19-47)."""
    _fields_ = [
        ("accumulationEnabled", C.c_uint32), ("numAccumulationFrames", C.c_uint32),
        ("numSpheres", C.c_uint32), ("numMeshes", C.c_uint32),
        ("numDirLights", C.c_uint32), ("numPointLights", C.c_uint32), ("numSpotLights", C.c_uint32),
        ("debugMode", C.c_uint32),
        ("textureDims", C.c_uint32 * 2), ("viewPlaneDims", C.c_float * 2),
        ("cameraInverseProjectionMatrix", C.c_float * 16), ("cameraInverseViewMatrix", C.c_float * 16),
        ("cameraPosition", C.c_float * 3), ("cameraNearZ", C.c_float),
        ("cameraUpDir", C.c_float * 3), ("dofStrength", C.c_float),
        ("cameraRightDir", C.c_float * 3), ("dofDistance", C.c_float),
        ("debugMaxCount", C.c_uint32), ("pad0", C.c_float * 3),
    ]


assert C.sizeof(RenderData) == 240


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "pixels", "segments", "n_oct", "n_mesh", "n_node", "n_tri", "mesh_hits", "sphere_hits",
        "tex_taps", "env_taps", "fast_nodes", "fast_tris")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


P = C.c_void_p
U32 = C.c_uint32
# name -> (restype, argtypes); every symbol include/okrt.h declares (minus okrt_create/get_api,
# which differ between product and oracle and are bound separately).
SIGNATURES = {
    "destroy": (C.c_int, [P]),
    "last_error": (C.c_char_p, [P]),
    "set_table": (C.c_int, [P, C.c_int, P, U32, U32]),
    "set_textures": (C.c_int, [P, P, U32, U32, U32]),
    "set_envmap": (C.c_int, [P, P, U32]),
    "scale_texture_rgba8": (C.c_int, [P, P, U32, U32, P, U32, U32]),
    "resize": (C.c_int, [P, U32, U32]),
    "set_tile": (C.c_int, [P, U32, U32, U32]),
    "tile_rows": (C.c_int, [P, C.POINTER(U32)]),
    "reset_accumulation": (C.c_int, [P]),
    "render": (C.c_int, [P, C.POINTER(RenderData)]),
    "render_frames": (C.c_int, [P, C.POINTER(RenderData), U32]),
    "synchronize": (C.c_int, [P]),
    "read_result_rgba8": (C.c_int, [P, P]),
    "read_accum_f32x4": (C.c_int, [P, P]),
    "set_option": (C.c_int, [P, C.c_int, C.c_int64]),
    "get_counters": (C.c_int, [P, C.POINTER(Counters)]),
    "reset_counters": (C.c_int, [P]),
}
# product-only entry points
PRODUCT_ONLY = {
    "abi_version": (C.c_int, []),
    "create": (C.c_int, [C.c_int, C.POINTER(P)]),
    "bind_framebuffers": (C.c_int, [P, P, P]),
    "set_stream": (C.c_int, [P, P]),
    "timer_begin": (C.c_int, [P]),
    "timer_end": (C.c_int, [P]),
    "timer_elapsed_ms": (C.c_int, [P, C.POINTER(C.c_float)]),
    "last_render_ms": (C.c_int, [P, C.POINTER(C.c_float)]),
    "get_stage_ms": (C.c_int, [P, C.POINTER(C.c_float * 3), C.POINTER(U32)]),
    "get_interval_ms": (C.c_int, [P, C.POINTER(C.c_float), U32, C.POINTER(U32), C.POINTER(U32), C.POINTER(U32)]),
    "write_accum_f32x4": (C.c_int, [P, P]),
    "get_api": (P, []),
    "commit": (C.c_int, [P]),
    "get_build_info": (C.c_int, [P, P]),
    "get_fast_structure": (C.c_int, [P, P, U32, P, U32]),
    "debug_kat": (C.c_int, [P, C.c_int, U32, P, U32, P, U32]),
    "comm_unique_id": (C.c_int, [P]),
    "comm_create": (C.c_int, [P, P, U32, U32, C.POINTER(P)]),
    "comm_create_all": (C.c_int, [C.POINTER(P), U32, C.POINTER(P)]),
    "comm_destroy": (C.c_int, [P]),
    "comm_last_error": (C.c_char_p, [P]),
    "mgpu_gather_accum": (C.c_int, [P, P, U32, P]),
    "mgpu_gather_accum_all": (C.c_int, [C.POINTER(P), C.POINTER(P), U32, U32, P]),
    "mgpu_read_gathered": (C.c_int, [P, P, P]),
    "mgpu_deinterleave": (C.c_int, [P, P, U32, U32, U32, P]),
    "device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "comm_transport": (C.c_char_p, [P]),
    "mgpu_reduce_accum": (C.c_int, [P, P, U32, P]),
    "mgpu_reduce_accum_all": (C.c_int, [C.POINTER(P), C.POINTER(P), U32, U32, P]),
    "peek_result_rgba8": (C.c_int, [P, P, C.POINTER(U32)]),
    "frame_size": (C.c_int, [P, C.POINTER(U32), C.POINTER(U32)]),
    "comm_exchange_ms": (C.c_int, [P, C.POINTER(C.c_float)]),
}
KAT_PCG_HASH, KAT_RANDOM_FLOAT, KAT_RAY_SPHERE, KAT_RAY_TRIANGLE, KAT_RAY_AABB, KAT_REFRACT, KAT_SAMPLE_TEXTURE, KAT_SAMPLE_ENV, KAT_CLOSEST_HIT, KAT_CLOSEST_HIT_FAST = range(10)
KAT_RANDOM_VECTOR, KAT_REFLECTANCE, KAT_REFLECT_DIRECTION, KAT_TRANSPARENCY_BOUNCE, KAT_PRIMARY_RAY = range(10, 15)
KAT_SEGMENT_RAY = 15
KAT_IEEE_SHORTCUTS = 16
KAT_CLOSEST_HIT_TRACE = 17
HEADER_SYMBOLS = ["okrt_" + n for n in list(SIGNATURES) + list(PRODUCT_ONLY)]


class OkrtError(RuntimeError):
    pass


class _Bound:
    """Functions of one library, reachable as attributes without their prefix."""

    def __init__(self, lib, prefix, table):
        self._lib = lib
        for name, (res, args) in table.items():
            fn = getattr(lib, prefix + name)  # AttributeError if the library lacks the symbol
            fn.restype = res
            fn.argtypes = args
            setattr(self, name, fn)


def bind(lib, prefix, extra=None):
    table = dict(SIGNATURES)
    if extra:
        table.update(extra)
    return _Bound(lib, prefix, table)


def device_library_path():
    # OKRT_LIBRARY: developer override to A/B another build of the same ABI
    return os.environ.get("OKRT_LIBRARY") or os.path.join(PKG_DIR, "libokrt.so")


_device = None


def load_device_library():
    """Load libokrt.so (the HIP product). Raises if it is not built; never falls back.

    In a process that also uses PyTorch, `import torch` FIRST: torch ships its own libamdhip64.so.7 and the dynamic
    loader then binds libokrt.so to that same runtime instance (needed to share device pointers / streams)."""
    global _device
    if _device is None:
        path = device_library_path()
        if not os.path.exists(path):
            raise OkrtError(f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()')")
        lib = C.CDLL(path)
        fns = bind(lib, "okrt_", PRODUCT_ONLY)
        if fns.abi_version() != OKRT_ABI_VERSION:
            raise OkrtError("libokrt.so ABI version mismatch")
        _device = fns
    return _device
