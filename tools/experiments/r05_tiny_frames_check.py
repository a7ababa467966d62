"""Frame-stacked packets on frames smaller than a tile and on frame counts around the group sizes: one okrt_render_frames against the same
number of okrt_render calls (bit for bit) and against the reference-order kernel. usage: python tools/experiments/r05_tiny_frames_check.py"""
import importlib, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.small_mesh_scene(seed=3, n_objects=5, detail=6, with_lights=False)
bad = 0
for (W, H) in ((1, 1), (5, 3), (8, 8), (9, 1), (1, 17), (7, 9), (16, 8), (33, 5)):
    for frames in (4, 8, 12, 19):
        rt, _, _ = scenes.instantiate(desc, W, H)
        rt.ctx.set_option(abi.OPT_MAX_BOUNCES, 2)
        rt.renderFrames(frames); a = rt.ctx.read_accum()
        rt.resetAccumulation()
        rt.ctx.set_option(abi.OPT_DEFER, 0)
        for _ in range(frames): rt.render()
        b = rt.ctx.read_accum()
        rt.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_REFERENCE)
        rt.resetAccumulation(); rt.renderFrames(frames); c = rt.ctx.read_accum()
        ok = np.array_equal(a, b, equal_nan=True) and (a[..., 3] == frames).all()
        same = ((a == c).all(-1) | (np.isnan(a).any(-1) & np.isnan(c).any(-1))).mean()
        if not ok or same < 0.9: bad += 1
        print(W, H, frames, ok, round(float(same), 3))
        rt.close()
print("bad", bad)
