"""The C-ABI shared library loads and exports every symbol include/okrt.h declares (no compute calls: this runs
on machines without a GPU), and the header's layout constants agree with the ctypes mirror."""
import ctypes as C
import os
import re

from conftest import REPO, abi, host


def header_symbols():
    text = open(os.path.join(REPO, "include", "okrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^(?:okrt_status|int|const char\*|const okrt_api\*)\s+(okrt_[a-z0-9_]+)\s*\(", text, flags=re.M)))


def test_libokrt_exports_every_declared_entry_point():
    lib = C.CDLL(abi.device_library_path())
    declared = header_symbols()
    assert len(declared) >= 27
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert set(abi.HEADER_SYMBOLS) <= set(declared)
    lib.okrt_abi_version.restype = C.c_int
    assert lib.okrt_abi_version() == abi.OKRT_ABI_VERSION


def api_table_entries():
    text = open(os.path.join(REPO, "include", "okrt.h")).read()
    body = text[text.index("typedef struct okrt_api {"):text.index("} okrt_api;")]
    return re.findall(r"\(\*([a-z0-9_]+)\)\(", body)


def test_api_table_is_complete(oracle):
    """ABI 2+: the okrt_api table carries the WHOLE surface; the product fills every entry, and every declared okrt_*
    entry point (bar create-time helpers that cannot go through a table) has one. ABI 3 appended six entries (device
    count, transport name, the samples-split reduce, peek, frame size) behind the 41 of ABI 2, ABI 4 one more (the root's
    exchange time): older hosts keep working."""
    names = api_table_entries()
    assert len(names) == 48 and len(set(names)) == len(names)
    assert names[41:] == ["device_count", "comm_transport", "mgpu_reduce_accum", "mgpu_reduce_accum_all", "peek_result_rgba8", "frame_size", "comm_exchange_ms"]
    lib = C.CDLL(abi.device_library_path())
    lib.okrt_get_api.restype = C.POINTER(C.c_void_p * (len(names) + 1))
    table = lib.okrt_get_api().contents
    assert C.cast(C.pointer(table), C.POINTER(C.c_uint32)).contents.value == abi.OKRT_ABI_VERSION
    assert all(table[i] for i in range(1, len(names) + 1))  # no NULL in the product's table
    declared = {s[len("okrt_"):] for s in header_symbols()} - {"abi_version", "get_api"}
    assert declared <= set(names), sorted(declared - set(names))
    # the CPU checker implements the first 19 (the scene-upload / render / read-back core) and leaves the rest NULL
    oracle.lib.okrt_oracle_get_api.restype = C.POINTER(C.c_void_p * (len(names) + 1))
    otable = oracle.lib.okrt_oracle_get_api().contents
    assert all(otable[i] for i in range(1, 20)) and not any(otable[i] for i in range(20, len(names) + 1))
    oracle.lib.okrt_oracle_get_api.restype = C.c_void_p


def test_layout_constants():
    assert C.sizeof(abi.RenderData) == 240
    offs = {n: getattr(abi.RenderData, n).offset for n, _ in abi.RenderData._fields_}
    # RaytracerCS.hlsl:19-47 / RayTracer.h:129-157
    assert offs["textureDims"] == 32 and offs["viewPlaneDims"] == 40 and offs["cameraInverseProjectionMatrix"] == 48
    assert offs["cameraInverseViewMatrix"] == 112 and offs["cameraPosition"] == 176 and offs["cameraNearZ"] == 188
    assert offs["cameraUpDir"] == 192 and offs["dofStrength"] == 204 and offs["cameraRightDir"] == 208
    assert offs["dofDistance"] == 220 and offs["debugMaxCount"] == 224
    assert C.sizeof(abi.Counters) == 96
    m = host.Material
    mo = {n: getattr(m, n).offset for n, _ in m._fields_}
    # GPU-Structs.hlsli:37-71
    assert (mo["albedo"], mo["albedoTextureId"], mo["roughness"], mo["normalMapIdx"], mo["emissionColour"], mo["emissionPower"],
            mo["transparency"], mo["indexOfRefraction"]) == (0, 12, 16, 40, 44, 56, 60, 64)
    d = m()
    assert (d.roughness, d.metallic, d.specular, d.transparency, d.indexOfRefraction, d.emissionPower) == (1.0, 0.0, 1.0, 0.0, 1.0, 0.0)
    assert d.albedoTextureId == d.normalMapIdx == 0xFFFFFFFF and list(d.emissionColour) == [1.0, 1.0, 1.0]


def test_product_does_not_link_or_load_the_oracle():
    import subprocess
    out = subprocess.run(["ldd", abi.device_library_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out
    out = subprocess.run(["ldd", host.host_library_path()], capture_output=True, text=True).stdout
    assert "oracle" not in out and "amdhip" not in out  # the host layer reaches the device only through okrt_api
    for root, _, files in os.walk(os.path.join(REPO, "gpu-raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) and f != "build.py":  # build.py only BUILDS the checker
                src = open(os.path.join(root, f)).read()
                assert "liboracle" not in src and "okrt_oracle_" not in src, f
