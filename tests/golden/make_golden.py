"""Generates the committed fixtures under tests/golden/ (run in the build container only).

  ref_host.npz       outputs of the REAL reference host code compiled in oracle/_ref (make -C oracle ref):
                     struct sizes, BvhBuilder trees for two tiny meshes, calculateProjectionData for three
                     camera poses, instance matrices for three transforms. Inputs are stored next to outputs.
  oracle_images.npz  float accumulations of the CPU oracle (oracle/liboracle.so) for small versions of the
                     benchmark configs. The reference ships no images that could pin these ("parity
                     unpinned"); they pin the oracle against regressions and give the GPU
                     tests a fixture that does not need the oracle binary.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import conftest  # noqa: E402

scenes, abi = conftest.scenes, conftest.abi
P = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def golden_scenes():
    """name -> (SceneDescription, W, H, frames, max_bounces); shared with the tests."""
    return {
        "spheres": (scenes.test_scene_spheres(), 64, 64, 1, 1),
        "small_lights": (scenes.small_mesh_scene(with_lights=True), 64, 36, 1, 1),
        "small_16spp": (scenes.small_mesh_scene(seed=9, with_lights=False), 64, 36, 16, 1),
        # rotated + translated instances; root-only octree: the reference refits octree CHILD boxes to the
        # LOCAL mesh boxes (quirk Q22, RayTracer.cpp:562), which wrongly culls moved meshes, so only the root
        # (built from transformed corners, :546-560) bounds them correctly
        "small_transforms": (scenes.small_mesh_scene(seed=4, transforms=True, octree=(0, 50)), 64, 36, 1, 1),
        "glass_b4": (scenes.glass_revolver_proxy(tex_size=64, env_face=32, detail=0.2), 64, 36, 1, 4),
        "dragon2k": (scenes.dragon_proxy(2000), 64, 36, 1, 1),
        "sponza_tiny": (scenes.sponza_proxy(tex_size=32, env_face=16, tri_scale=0.02), 64, 36, 1, 1),
    }


def tiny_meshes():
    rng = np.random.default_rng(42)
    c = rng.uniform(-4, 4, (96, 1, 3)).astype(np.float32)
    soup = (c + rng.normal(0, 0.4, (96, 3, 3))).astype(np.float32)
    grid, _ = scenes.grid_surface(lambda U, V: np.stack([U * 3, np.sin(U * 5) * np.cos(V * 4), V * 2], -1), 8, 6)
    return {"soup96": (soup, 5, 30), "grid96": (grid, 3, 6), "soup96_leaf1": (soup, 1, 100)}


def main():
    ref = C.CDLL(os.path.join(REPO, "oracle", "_ref", "libref_host.so"))
    out = {}
    sizes = np.zeros(8, np.uint32)
    ref.ref_struct_sizes(P(sizes))
    out["struct_sizes"] = sizes
    ref.ref_bvh_build.restype = C.c_uint32
    for name, (tris, leaf, depth) in tiny_meshes().items():
        n = ref.ref_bvh_build(P(tris), len(tris), leaf, depth, None, 0, None)
        nodes = np.zeros((n, 9), np.uint32)
        order = np.zeros(len(tris), np.uint32)
        ref.ref_bvh_build(P(tris), len(tris), leaf, depth, P(nodes), n, P(order))
        out[f"bvh_{name}_tris"] = tris
        out[f"bvh_{name}_params"] = np.array([leaf, depth], np.uint32)
        out[f"bvh_{name}_nodes"] = nodes
        out[f"bvh_{name}_order"] = order
    cams = np.array([[60, 0, 0, 0, -90, 0, 90, 0.1, 1000, 1920, 1080],       # Application.cpp:67-70
                     [1.2, 4.1, 10.1, 19.4, -10.9, 0, 90, 0.1, 1000, 256, 256],
                     [-3, 7, 2, -35, 140, 12, 55, 0.5, 300, 1280, 720]], np.float32)
    cam_out = np.zeros((len(cams), 48), np.float32)
    ref.ref_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, C.c_void_p]
    for i, cpar in enumerate(cams):
        pos, rot = cpar[0:3].copy(), cpar[3:6].copy()
        ref.ref_camera(P(pos), P(rot), float(cpar[6]), float(cpar[7]), float(cpar[8]), int(cpar[9]), int(cpar[10]), P(cam_out[i]))
    out["camera_in"], out["camera_out"] = cams, cam_out
    xf = np.array([[0, 0, 0, 0, 0, 0, 1, 1, 1], [3, -2, 5, 30, 45, -60, 1, 1, 1], [-1, 4, 2, 10, 200, 5, 2, 0.5, 1.5]], np.float32)
    xm = np.zeros((len(xf), 32), np.float32)
    for i, t in enumerate(xf):
        pos, rot, sc = t[0:3].copy(), t[3:6].copy(), t[6:9].copy()
        m, inv = np.zeros(16, np.float32), np.zeros(16, np.float32)
        ref.ref_instance_matrices(P(pos), P(rot), P(sc), P(m), P(inv))
        xm[i, :16], xm[i, 16:] = m, inv
    out["xform_in"], out["xform_out"] = xf, xm
    np.savez_compressed(os.path.join(HERE, "ref_host.npz"), **out)

    oracle = conftest.Oracle()
    imgs = {}
    for name, (desc, w, h, frames, mb) in golden_scenes().items():
        rt, _, _ = oracle.instantiate(desc, w, h)
        rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
        rt.ctx.set_option(abi.OPT_COUNTERS, 1)
        for _ in range(frames):
            rt.render()
        imgs[name] = rt.ctx.read_accum()
        c = rt.ctx.counters()
        imgs[name + "_counters"] = np.array([c[k] for k in sorted(c)], np.uint64)
        bb, tri = oracle.counter_images(rt.ctx)
        imgs[name + "_bb"] = bb.astype(np.uint16) if bb.max() < 65536 else bb
        rt.close()
        print(name, imgs[name].shape, float(np.nanmean(imgs[name][..., :3])))
    np.savez_compressed(os.path.join(HERE, "oracle_images.npz"), **imgs)
    for f in ("ref_host.npz", "oracle_images.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
