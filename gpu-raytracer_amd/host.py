"""Python view of the host side of the drop-in.

* `Context`   — one okrt device context (include/okrt.h) with numpy in/out.
* `ResourceManager`, `Scene`, `RayTracer` — ctypes wrappers over the C++ host layer
  (gpu-raytracer_amd/host, libokrt_host.so), which mirrors the reference's `RayTracer` driver
  (source/Graphics/RayTracer.{h,cpp}): same method names, same argument meaning.

The host layer binds to a device layer through an `okrt_api` table; `RayTracer(api=None)` uses the
HIP product (libokrt.so). Tests may pass the oracle's table instead.
"""
import ctypes as C
import os

import numpy as np

from . import _abi
from ._abi import Counters, OkrtError, RenderData

P, U32 = C.c_void_p, C.c_uint32


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Context:
    """An okrt_ctx. `fns` is a bound function set (product by default)."""

    def __init__(self, device=0, fns=None, handle=None):
        self.fns = fns or _abi.load_device_library()
        self._own = handle is None
        if handle is None:
            h = P()
            st = self.fns.create(device, C.byref(h))
            if st != 0:
                raise OkrtError(f"okrt_create(device={device}) failed with status {st}")
            handle = h
        self.h = handle
        self.W = self.H = 0

    def _ck(self, st, what):
        if st != 0:
            raise OkrtError(f"{what}: status {st}: {self.fns.last_error(self.h).decode()}")

    def close(self):
        if self._own and self.h:
            self.fns.destroy(self.h)
        self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_table(self, slot, array):
        if array is None or len(array) == 0:
            self._ck(self.fns.set_table(self.h, slot, None, _abi.TABLE_STRIDE[slot], 0), "set_table")
            return
        a = np.ascontiguousarray(array)
        stride = _abi.TABLE_STRIDE[slot]
        assert a.nbytes % stride == 0, (slot, a.nbytes, stride)
        self._ck(self.fns.set_table(self.h, slot, _ptr(a), stride, a.nbytes // stride), "set_table")

    def set_table_raw(self, slot, array, elem_bytes, count):
        a = np.ascontiguousarray(array)
        return self.fns.set_table(self.h, slot, _ptr(a), elem_bytes, count)

    def set_textures(self, rgba8):
        if rgba8 is None:
            self._ck(self.fns.set_textures(self.h, None, 0, 0, 0), "set_textures")
            return
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        n, h, w, ch = a.shape
        assert ch == 4
        self._ck(self.fns.set_textures(self.h, _ptr(a), n, w, h), "set_textures")

    def set_envmap(self, faces):
        if faces is None:
            self._ck(self.fns.set_envmap(self.h, None, 0), "set_envmap")
            return
        a = np.ascontiguousarray(faces, dtype=np.uint8)
        assert a.shape[0] == 6 and a.shape[1] == a.shape[2] and a.shape[3] == 4
        self._ck(self.fns.set_envmap(self.h, _ptr(a), a.shape[1]), "set_envmap")

    def scale_texture(self, rgba8, new_w, new_h):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        out = np.empty((new_h, new_w, 4), dtype=np.uint8)
        self._ck(self.fns.scale_texture_rgba8(self.h, _ptr(a), a.shape[1], a.shape[0], _ptr(out), new_w, new_h), "scale_texture")
        return out

    def resize(self, w, h):
        self._ck(self.fns.resize(self.h, w, h), "resize")
        self.W, self.H = w, h

    def set_tile(self, stripe_rows, stripe_index, stripe_stride):
        self._ck(self.fns.set_tile(self.h, stripe_rows, stripe_index, stripe_stride), "set_tile")

    def tile_rows(self):
        r = U32()
        self._ck(self.fns.tile_rows(self.h, C.byref(r)), "tile_rows")
        return r.value

    def reset_accumulation(self):
        self._ck(self.fns.reset_accumulation(self.h), "reset_accumulation")

    def render(self, rd):
        self._ck(self.fns.render(self.h, C.byref(rd)), "render")

    def render_frames(self, rd, frames):
        self._ck(self.fns.render_frames(self.h, C.byref(rd), frames), "render_frames")

    def synchronize(self):
        self._ck(self.fns.synchronize(self.h), "synchronize")

    def read_accum(self):
        out = np.empty((self.tile_rows(), self.W, 4), dtype=np.float32)
        self._ck(self.fns.read_accum_f32x4(self.h, _ptr(out)), "read_accum")
        return out

    def write_accum(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        assert a.shape == (self.tile_rows(), self.W, 4)
        self._ck(self.fns.write_accum_f32x4(self.h, _ptr(a)), "write_accum")

    def read_result(self):
        out = np.empty((self.tile_rows(), self.W, 4), dtype=np.uint8)
        self._ck(self.fns.read_result_rgba8(self.h, _ptr(out)), "read_result")
        return out

    def peek_result(self):
        """(rgba8 picture, numAccumulationFrames it shows): okrt_peek_result_rgba8 — no flush of deferred frames."""
        out = np.zeros((self.tile_rows(), self.W, 4), dtype=np.uint8)
        n = U32()
        self._ck(self.fns.peek_result_rgba8(self.h, _ptr(out), C.byref(n)), "peek_result")
        return out, n.value

    def frame_size(self):
        w, h = U32(), U32()
        self._ck(self.fns.frame_size(self.h, C.byref(w), C.byref(h)), "frame_size")
        return w.value, h.value

    def set_option(self, opt, value):
        self._ck(self.fns.set_option(self.h, opt, int(value)), "set_option")

    def counters(self):
        c = Counters()
        self._ck(self.fns.get_counters(self.h, C.byref(c)), "get_counters")
        return c.as_dict()

    def reset_counters(self):
        self._ck(self.fns.reset_counters(self.h), "reset_counters")

    # product-only helpers -----------------------------------------------------------------
    def timer_begin(self):
        self._ck(self.fns.timer_begin(self.h), "timer_begin")

    def timer_end(self):
        self._ck(self.fns.timer_end(self.h), "timer_end")

    def timer_elapsed_ms(self):
        ms = C.c_float()
        self._ck(self.fns.timer_elapsed_ms(self.h, C.byref(ms)), "timer_elapsed_ms")
        return ms.value

    def stage_ms(self):
        """(trace_ms, shade_ms, other_ms, frames) summed over the OKRT_OPT_TIMING ring."""
        ms = (C.c_float * 3)()
        n = U32()
        self._ck(self.fns.get_stage_ms(self.h, C.byref(ms), C.byref(n)), "get_stage_ms")
        return ms[0], ms[1], ms[2], n.value

    def interval_ms(self):
        """([ms per kernel interval, summed over the ring's launch sets], launch sets, frames): okrt_get_interval_ms."""
        ms = (C.c_float * 160)()
        n, launches, frames = U32(), U32(), U32()
        self._ck(self.fns.get_interval_ms(self.h, ms, 160, C.byref(n), C.byref(launches), C.byref(frames)), "get_interval_ms")
        return [ms[k] for k in range(min(n.value, 160))], launches.value, frames.value

    def bind_framebuffers(self, accum_ptr, result_ptr):
        self._ck(self.fns.bind_framebuffers(self.h, accum_ptr, result_ptr), "bind_framebuffers")

    def set_stream(self, stream_ptr):
        self._ck(self.fns.set_stream(self.h, stream_ptr), "set_stream")

    def kat(self, op, records, out_floats):
        """Run one device leaf function over `records` ([n, k] float32; integers as bit patterns): okrt_debug_kat."""
        a = np.ascontiguousarray(records, dtype=np.float32)
        out = np.zeros((a.shape[0], out_floats), np.float32)
        self._ck(self.fns.debug_kat(self.h, op, a.shape[0], a.ctypes.data_as(P), a.shape[1], out.ctypes.data_as(P), out_floats), "debug_kat")
        return out

    def commit(self):
        """Validate the tables and (re)build the traversal structure now (okrt_commit)."""
        self._ck(self.fns.commit(self.h), "commit")

    def build_info(self):
        class _Info(C.Structure):
            _fields_ = [("builder", U32), ("num_nodes", U32), ("num_triangles", U32), ("max_depth", U32),
                        ("device_ms", C.c_float), ("wall_ms", C.c_float), ("flags", U32)]
        bi = _Info()
        self._ck(self.fns.get_build_info(self.h, C.byref(bi)), "get_build_info")
        return {k: getattr(bi, k) for k, _ in _Info._fields_}

    def fast_structure(self):
        """(nodes uint32 [n,16], tris uint32 [m,12]) — the derived structure, layouts of csrc/okrt_fast.h."""
        bi = self.build_info()
        nodes = np.zeros((max(bi["num_nodes"], 1), 16), np.uint32)
        tris = np.zeros((max(bi["num_triangles"], 1), 12), np.uint32)
        self._ck(self.fns.get_fast_structure(self.h, nodes.ctypes.data_as(P), nodes.shape[0], tris.ctypes.data_as(P), tris.shape[0]),
                 "get_fast_structure")
        return nodes[:bi["num_nodes"]], tris[:bi["num_triangles"]]


# ---- C++ host layer -----------------------------------------------------------------------------
class Material(C.Structure):
    """68-byte Material (GPU-Structs.hlsli:37-71); defaults of Components.h:42-56."""
    _fields_ = [
        ("albedo", C.c_float * 3), ("albedoTextureId", U32),
        ("roughness", C.c_float), ("roughnessTextureId", U32),
        ("metallic", C.c_float), ("metallicTextureId", U32),
        ("specular", C.c_float), ("specularTextureId", U32),
        ("normalMapIdx", U32),
        ("emissionColour", C.c_float * 3), ("emissionPower", C.c_float),
        ("transparency", C.c_float), ("indexOfRefraction", C.c_float),
    ]

    def __init__(self, **kw):
        super().__init__()
        host_library().okh_material_default(C.byref(self))
        for k, v in kw.items():
            if k in ("albedo", "emissionColour"):
                setattr(self, k, (C.c_float * 3)(*v))
            else:
                setattr(self, k, v)


assert C.sizeof(Material) == 68

_host = None


def host_library_path():
    # OKRT_HOST_LIBRARY: developer override (the -fsanitize build of `make asan`)
    return os.environ.get("OKRT_HOST_LIBRARY") or os.path.join(_abi.PKG_DIR, "libokrt_host.so")


def host_library():
    global _host
    if _host is None:
        path = host_library_path()
        if not os.path.exists(path):
            raise OkrtError(f"{path} is missing: build it first (__graft_entry__.build())")
        lib = C.CDLL(path)
        F3 = C.POINTER(C.c_float)
        sig = {
            "okh_material_default": (None, [P]),
            "okh_resources_create": (P, []), "okh_resources_destroy": (None, [P]),
            "okh_resources_add_mesh": (U32, [P, P, P, U32]), "okh_resources_add_texture": (U32, [P, P, U32, U32]),
            "okh_resources_import_obj": (C.c_int, [P, C.c_char_p, C.c_float, C.c_char_p, U32]),
            "okh_scene_create": (P, []), "okh_scene_destroy": (None, [P]), "okh_scene_create_entity": (U32, [P]),
            "okh_scene_set_transform": (None, [P, U32, F3, F3, F3]),
            "okh_scene_add_mesh": (None, [P, U32, U32, P]), "okh_scene_add_sphere": (None, [P, U32, C.c_float, P]),
            "okh_scene_add_dir_light": (None, [P, U32, F3, C.c_float, C.c_float]),
            "okh_scene_add_point_light": (None, [P, U32, F3, C.c_float, C.c_float]),
            "okh_scene_add_spot_light": (None, [P, U32, F3, C.c_float, C.c_float, C.c_float]),
            "okh_scene_add_camera": (None, [P, U32, C.c_float, C.c_float, C.c_float]),
            "okh_raytracer_create": (P, [P, C.c_int]), "okh_raytracer_destroy": (None, [P]),
            "okh_raytracer_last_error": (C.c_char_p, [P]),
            "okh_raytracer_initiate": (C.c_int, [P, U32, U32, P, P, U32, U32]),
            "okh_raytracer_loadMeshAndBvhData": (C.c_int, [P, U32, U32]),
            "okh_raytracer_createOctTree": (C.c_int, [P, P, U32, U32]),
            "okh_raytracer_setScene": (None, [P, P]), "okh_raytracer_render": (C.c_int, [P]),
            "okh_raytracer_renderFrames": (C.c_int, [P, U32]),
            "okh_raytracer_toggleAccumulation": (None, [P, C.c_int]), "okh_raytracer_resetAccumulation": (None, [P]),
            "okh_raytracer_getNumAccumulationFrames": (U32, [P]),
            "okh_raytracer_setDOF": (None, [P, C.c_float, C.c_float]), "okh_raytracer_onResize": (C.c_int, [P, U32, U32]),
            "okh_raytracer_setDebugMode": (None, [P, U32, U32]),
            "okh_raytracer_device_ctx": (P, [P]), "okh_raytracer_render_data": (C.POINTER(RenderData), [P]),
            "okh_raytracer_calculateProjectionData": (None, [P]), "okh_raytracer_updateBuffers": (C.c_int, [P]),
            "okh_raytracer_table": (C.c_int, [P, U32, C.POINTER(P), C.POINTER(U32), C.POINTER(U32)]),
            "okh_bvh_build": (U32, [P, U32, U32, U32, P, U32, P]),
            "okh_instance_matrices": (None, [F3, F3, F3, F3, F3]),
            "okh_cut_environment_cross": (U32, [P, U32, U32, P, U32]),
            "okh_write_png": (C.c_int, [C.c_char_p, U32, U32, P]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _host = lib
    return _host


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class ResourceManager:
    def __init__(self):
        self.lib = host_library()
        self.h = self.lib.okh_resources_create()

    def add_mesh(self, positions, info=None):
        """positions: [N,3,3] float32; info: [N,33] float32 (TriangleInfo) or None. Returns meshID."""
        p = np.ascontiguousarray(positions, dtype=np.float32)
        i = np.ascontiguousarray(info, dtype=np.float32) if info is not None else None
        n = p.size // 9
        if i is not None:
            assert i.size == n * 33
        return self.lib.okh_resources_add_mesh(self.h, _ptr(p), _ptr(i), n)

    def add_texture(self, rgba8):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        return self.lib.okh_resources_add_texture(self.h, _ptr(a), a.shape[1], a.shape[0])

    def import_obj(self, path, scale=1.0):
        """ResourceManager::importAssets for OBJ/MTL: returns [{name, meshId, numTris, albedo, roughness, metallic,
        specular, normal}] (texture entries are the file names of the .mtl; decode + add_texture them yourself)."""
        import json
        buf = C.create_string_buffer(1 << 22)
        n = self.lib.okh_resources_import_obj(self.h, os.fsencode(path), float(scale), buf, len(buf))
        if n < 0:
            raise OkrtError("import_obj: " + buf.value.decode(errors="replace"))
        return json.loads(buf.value.decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.okh_resources_destroy(self.h)
            self.h = None


class Scene:
    def __init__(self):
        self.lib = host_library()
        self.h = self.lib.okh_scene_create()

    def create_entity(self, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        e = self.lib.okh_scene_create_entity(self.h)
        self.lib.okh_scene_set_transform(self.h, e, _f3(position), _f3(rotation), _f3(scale))
        return e

    def set_transform(self, e, position=(0, 0, 0), rotation=(0, 0, 0), scale=(1, 1, 1)):
        """Move an entity (the reference's ImGui transform edit, Application.cpp:287-317); follow with createOctTree."""
        self.lib.okh_scene_set_transform(self.h, e, _f3(position), _f3(rotation), _f3(scale))

    def add_mesh(self, e, mesh_id, material=None):
        m = material or Material()
        self.lib.okh_scene_add_mesh(self.h, e, mesh_id, C.byref(m))

    def add_sphere(self, e, radius, material=None):
        m = material or Material()
        self.lib.okh_scene_add_sphere(self.h, e, radius, C.byref(m))

    def add_dir_light(self, e, colour=(1, 1, 1), intensity=1.0, effective_angle=2.0):
        self.lib.okh_scene_add_dir_light(self.h, e, _f3(colour), intensity, effective_angle)

    def add_point_light(self, e, colour=(1, 1, 1), intensity=1.0, radius=1.0):
        self.lib.okh_scene_add_point_light(self.h, e, _f3(colour), intensity, radius)

    def add_spot_light(self, e, colour=(1, 1, 1), intensity=1.0, radius=1.0, max_angle=90.0):
        self.lib.okh_scene_add_spot_light(self.h, e, _f3(colour), intensity, radius, max_angle)

    def add_camera(self, e, fov=90.0, near_z=0.1, far_z=1000.0):
        self.lib.okh_scene_add_camera(self.h, e, fov, near_z, far_z)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.okh_scene_destroy(self.h)
            self.h = None


class RayTracer:
    """Mirror of the reference's RayTracer (source/Graphics/RayTracer.h:82-116) over the C ABI."""

    def __init__(self, device=0, api=None, fns=None):
        self.lib = host_library()
        if api is None:
            fns = _abi.load_device_library()
            api = fns.get_api()
        self.fns = fns
        self.h = self.lib.okh_raytracer_create(api, device)
        ctx = self.lib.okh_raytracer_device_ctx(self.h)
        if not ctx:
            msg = self.lib.okh_raytracer_last_error(self.h).decode()
            self.lib.okh_raytracer_destroy(self.h)
            self.h = None
            raise OkrtError("RayTracer: " + msg)
        self.ctx = Context(fns=fns, handle=P(ctx))
        self._keep = []

    def _ck(self, ok, what):
        if not ok:
            raise OkrtError(f"{what}: {self.lib.okh_raytracer_last_error(self.h).decode()}")

    def initiate(self, width, height, resources, env_cross=None):
        self._keep.append(resources)
        env = np.ascontiguousarray(env_cross, dtype=np.uint8) if env_cross is not None else None
        ew, eh = (env.shape[1], env.shape[0]) if env is not None else (0, 0)
        self._ck(self.lib.okh_raytracer_initiate(self.h, width, height, resources.h, _ptr(env), ew, eh), "initiate")
        self.ctx.W, self.ctx.H = width, height

    def loadMeshAndBvhData(self, max_depth, max_leaf_triangles):
        self._ck(self.lib.okh_raytracer_loadMeshAndBvhData(self.h, max_depth, max_leaf_triangles), "loadMeshAndBvhData")

    def createOctTree(self, scene, max_depth, max_leaf_objects):
        self._keep.append(scene)
        self._ck(self.lib.okh_raytracer_createOctTree(self.h, scene.h, max_depth, max_leaf_objects), "createOctTree")

    def setScene(self, scene):
        self._keep.append(scene)
        self.lib.okh_raytracer_setScene(self.h, scene.h)

    def render(self):
        self._ck(self.lib.okh_raytracer_render(self.h), "render")

    def renderFrames(self, frames):
        self._ck(self.lib.okh_raytracer_renderFrames(self.h, frames), "renderFrames")

    def toggleAccumulation(self, enable):
        self.lib.okh_raytracer_toggleAccumulation(self.h, int(bool(enable)))

    def resetAccumulation(self):
        self.lib.okh_raytracer_resetAccumulation(self.h)

    def getNumAccumulationFrames(self):
        return self.lib.okh_raytracer_getNumAccumulationFrames(self.h)

    def setDOF(self, strength, distance):
        self.lib.okh_raytracer_setDOF(self.h, strength, distance)

    def onResize(self, width, height):
        self._ck(self.lib.okh_raytracer_onResize(self.h, width, height), "onResize")
        self.ctx.W, self.ctx.H = width, height

    def setDebugMode(self, mode, max_count=500):
        self.lib.okh_raytracer_setDebugMode(self.h, mode, max_count)

    # harness access ---------------------------------------------------------------------------
    @property
    def render_data(self):
        return self.lib.okh_raytracer_render_data(self.h).contents

    def calculateProjectionData(self):
        self.lib.okh_raytracer_calculateProjectionData(self.h)

    def updateBuffers(self):
        self._ck(self.lib.okh_raytracer_updateBuffers(self.h), "updateBuffers")

    def table(self, which):
        """Host copy of a table the driver built, as a uint8 array [count, stride]."""
        data, count, elem = P(), U32(), U32()
        if not self.lib.okh_raytracer_table(self.h, which, C.byref(data), C.byref(count), C.byref(elem)):
            raise OkrtError(f"no such table {which}")
        if which == 3:  # scaled texture array: count = bytes, elem = side
            side = elem.value
            if not count.value:
                return np.zeros((0, 0, 0, 4), np.uint8)
            buf = (C.c_uint8 * count.value).from_address(data.value)
            return np.frombuffer(buf, dtype=np.uint8).reshape(-1, side, side, 4).copy()
        n = count.value * elem.value
        if n == 0:
            return np.zeros((0, elem.value), np.uint8)
        buf = (C.c_uint8 * n).from_address(data.value)
        return np.frombuffer(buf, dtype=np.uint8).reshape(count.value, elem.value).copy()

    def close(self):
        if self.h:
            self.lib.okh_raytracer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def write_png(path, rgba8):
    """Application::saveScreenshot's stbi_write_png through the host layer's own encoder (host/png_writer.cpp)."""
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    assert a.ndim == 3 and a.shape[2] == 4
    if not host_library().okh_write_png(os.fsencode(path), a.shape[1], a.shape[0], _ptr(a)):
        raise OkrtError(f"cannot write {path}")


def bvh_build(tris, max_leaf_triangles, max_depth):
    """Standalone per-mesh build: returns (nodes[n,9] uint32 view, tri_order[N])."""
    lib = host_library()
    t = np.ascontiguousarray(tris, dtype=np.float32)
    n = t.size // 9
    cnt = lib.okh_bvh_build(_ptr(t), n, max_leaf_triangles, max_depth, None, 0, None)
    nodes = np.zeros((cnt, 9), dtype=np.uint32)
    order = np.zeros(max(n, 1), dtype=np.uint32)
    lib.okh_bvh_build(_ptr(t), n, max_leaf_triangles, max_depth, _ptr(nodes), cnt, _ptr(order))
    return nodes, order[:n]
