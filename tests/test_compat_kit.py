"""SURVEY §8f rank 3, the link-level drop-in: gpu-raytracer_amd/host/compat provides the reference's RayTracer /
GPUStorage / RenderTexture / Okay:: names over include/okrt.h. This test compiles tests/compat/application_stub.cpp — the calls
of Application.cpp:25-52,61-128,131-152,312-317,408-411,661-674 — against that directory AND the reference's own
Scene / Entity / Components / ResourceManager sources where they lie (build container only: the reference checkout never
travels, so the test skips elsewhere), runs it on the CPU checker's okrt_api table, and requires render.png to equal, byte
for byte, the image the Python binding of the same host layer produces for the same OBJ scene."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, abi, host, scenes

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "source", "Scene")), reason="needs the reference checkout (build container only)")

OBJ = """mtllib scene.mtl
o floor
v -6 0 -6
v 6 0 -6
v 6 0 6
v -6 0 6
vt 0 0
vt 3 0
vt 3 3
vt 0 3
vn 0 1 0
usemtl tiles
f 1/1/1 2/2/1 3/3/1 4/4/1
o pyramid
v -1.5 0 -1.5
v 1.5 0 -1.5
v 1.5 0 1.5
v -1.5 0 1.5
v 0 2.5 0
usemtl plain
f 5 6 9
f 6 7 9
f 7 8 9
f 8 5 9
"""
MTL = """newmtl tiles
Kd 1 1 1
map_Kd tiles.png
map_Ks spec.png
newmtl plain
Kd 0.8 0.3 0.2
"""


def build_stub(tmp_path):
    compat = os.path.join(REPO, "gpu-raytracer_amd", "host", "compat")
    pkg = os.path.join(REPO, "gpu-raytracer_amd")
    exe = str(tmp_path / "application_stub")
    srcs = [os.path.join(REPO, "tests", "compat", "application_stub.cpp"),
            os.path.join(compat, "DirectX", "DX11.cpp"), os.path.join(compat, "DirectX", "RenderTexture.cpp"),
            os.path.join(compat, "Graphics", "GPUStorage.cpp"), os.path.join(compat, "Graphics", "RayTracer.cpp"),
            os.path.join(compat, "Graphics", "Importer.cpp"),
            os.path.join(REF, "source", "Scene", "Scene.cpp"), os.path.join(REF, "source", "Graphics", "ResourceManager.cpp")]
    cmd = ["g++", "-O1", "-std=c++20", "-DDIST", "-DGLM_FORCE_XYZW_ONLY", "-include", "cfloat", "-include", "cstring", "-include", "string", "-w",
           "-ffp-contract=off", "-I" + compat, "-I" + os.path.join(REF, "source"), "-I" + os.path.join(REF, "deps", "include"),
           "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(pkg, "host")] + srcs + [
           "-L" + pkg, "-lokrt_host", "-lokrt", "-L/opt/rocm/lib", "-lamdhip64", "-ldl", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stderr[-4000:]
    return exe


def test_application_calls_compile_link_and_render_like_the_python_binding(tmp_path, oracle):
    from PIL import Image
    (tmp_path / "scene.obj").write_text(OBJ)
    (tmp_path / "scene.mtl").write_text(MTL)
    rng = np.random.default_rng(4)
    Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)).resize((64, 64), Image.NEAREST).save(tmp_path / "tiles.png")
    Image.fromarray(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)).save(tmp_path / "spec.png")
    cross = scenes.sky_cross(16)
    Image.fromarray(cross).save(tmp_path / "cross.png")
    W, H, frames = 160, 90, 6
    exe = build_stub(tmp_path)
    out = tmp_path / "render.png"
    run = subprocess.run([exe, os.path.join(REPO, "oracle", "liboracle.so"), "okrt_oracle_get_api", str(tmp_path / "scene.obj"), str(tmp_path / "cross.png"),
                          str(W), str(H), str(frames), str(out)], capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr + run.stdout
    assert f"frames {frames} meshes 2 triangles 6 octree nodes" in run.stdout
    got = np.asarray(Image.open(out))
    assert got.shape == (H, W, 4)

    # the same scene and call sequence through the Python binding of the host layer, on the same device layer
    res, sc = host.ResourceManager(), host.Scene()
    scenes.load_meshes_as_entities(res, sc, str(tmp_path / "scene.obj"), "", 1.0)
    sc.add_camera(sc.create_entity((0.0, 2.0, 7.0), (10.0, 180.0, 0.0)), 80.0, 0.1, 1000.0)
    sc.add_point_light(sc.create_entity((2.0, 6.0, 3.0)), (1, 1, 1), 2.0, 1.0)
    sc.add_sphere(sc.create_entity((-2.5, 0.8, 1.0)), 0.8, host.Material(albedo=(0.9, 0.2, 0.2)))
    rt = host.RayTracer(api=oracle.api, fns=oracle.fns)
    rt.initiate(W, H, res, np.asarray(Image.open(tmp_path / "cross.png").convert("RGBA")))
    rt.setScene(sc)
    rt.createOctTree(sc, 2, 50)
    for _ in range(frames // 2):
        rt.render()
    rt.loadMeshAndBvhData(8, 2)
    rt.resetAccumulation()
    rt.createOctTree(sc, 2, 50)
    rt.setDOF(0.05, 6.0)
    for _ in range(frames):
        rt.render()
    want = rt.ctx.read_result()
    rt.close()
    assert want[..., :3].std() > 10
    assert np.array_equal(got, want)
