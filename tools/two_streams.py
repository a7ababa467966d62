"""Experiment: do two contexts (two HIP streams) of one process overlap on the GPU? Each owns every other 8-row stripe
of the 1080p Sponza-class frame; compare with one context rendering the whole frame."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.sponza_proxy()
W, H, F, SETS = 1920, 1080, 64, 4
def mk(tile=None):
    rt, sc, res = scenes.instantiate(desc, W, H)
    if tile: rt.ctx.set_tile(*tile)
    rt.renderFrames(F); rt.ctx.synchronize()
    return rt
one = mk()
t0 = time.perf_counter()
for _ in range(SETS): one.renderFrames(F)
one.ctx.synchronize()
print(f"one context, full frame: {(time.perf_counter()-t0)*1e3/(SETS*F):.4f} ms/frame")
one.close()
a, b = mk((8, 0, 2)), mk((8, 1, 2))
for blocks in (None,):
    t0 = time.perf_counter()
    for _ in range(SETS):
        a.renderFrames(F); b.renderFrames(F)
    a.ctx.synchronize(); b.ctx.synchronize()
    print(f"two contexts (two streams), half a frame each: {(time.perf_counter()-t0)*1e3/(SETS*F):.4f} ms/frame")
t0 = time.perf_counter()
for _ in range(SETS):
    a.renderFrames(F); a.ctx.synchronize(); b.renderFrames(F); b.ctx.synchronize()
print(f"the same, serialised: {(time.perf_counter()-t0)*1e3/(SETS*F):.4f} ms/frame")
