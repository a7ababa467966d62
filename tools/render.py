#!/usr/bin/env python3
"""Headless stand-in for the reference's window + "Save screenshot" (Application.cpp:661-674): render a scene for N
frames on the GPU and write render.png from the RGBA8 result buffer.

  python tools/render.py --scene sponza|dragon|dragon80k|glass|spheres|small  [--obj file.obj --textures dir --scale s]
                         [--size 1920x1080] [--frames 64] [--bounces 1] [--out render.png] [--device 0]
"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("gpu-raytracer_amd._abi")
host = importlib.import_module("gpu-raytracer_amd.host")
scenes = importlib.import_module("gpu-raytracer_amd.scenes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="sponza")
    ap.add_argument("--obj")
    ap.add_argument("--textures", default="")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--camera", default="", help="px,py,pz,rx,ry,rz,fov (degrees)")
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=1)
    ap.add_argument("--out", default="render.png")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.lower().split("x"))
    if a.obj:  # Application.cpp:44: loadMeshesAsEntities(path, texturePath, scale)
        res, scene = host.ResourceManager(), host.Scene()
        scenes.load_meshes_as_entities(res, scene, a.obj, a.textures, a.scale)
        cam = [float(v) for v in a.camera.split(",")] if a.camera else [60, 0, 0, 0, -90, 0, 90]  # Application.cpp:67-70
        scene.add_camera(scene.create_entity(cam[0:3], cam[3:6]), cam[6], 0.1, 1000.0)
        rt = host.RayTracer(device=a.device)
        rt.initiate(W, H, res, scenes.sky_cross(512))
        rt.setScene(scene)
        rt.createOctTree(scene, 2, 50)
    else:
        build = {"sponza": scenes.sponza_proxy, "dragon": lambda: scenes.dragon_proxy(871414), "dragon80k": lambda: scenes.dragon_proxy(80000),
                 "glass": scenes.glass_revolver_proxy, "spheres": scenes.test_scene_spheres, "small": scenes.small_mesh_scene}[a.scene]
        rt, scene, res = scenes.instantiate(build(), W, H, device=a.device)
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, a.bounces)
    rt.renderFrames(a.frames)
    host.write_png(a.out, rt.ctx.read_result())  # RGBA8, row 0 = top (stbi_write_png of the result target)
    print(f"{a.out}: {W}x{H}, {rt.getNumAccumulationFrames()} frames")


if __name__ == "__main__":
    main()
