"""Times okrt_read_result_rgba8 / okrt_peek_result_rgba8 / okrt_read_accum_f32x4 at 1080p (what a presenting host pays per frame)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
abi = importlib.import_module("gpu-raytracer_amd._abi")
scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.small_mesh_scene(seed=4, n_objects=5, detail=10)
rt, sc, res = scenes.instantiate(desc, 1920, 1080)
rt.renderFrames(16)
rt.ctx.synchronize()
for name, fn in (("read_result", rt.ctx.read_result), ("peek_result", rt.ctx.peek_result), ("read_accum", rt.ctx.read_accum)):
    ts = []
    for _ in range(6):
        t = time.perf_counter(); fn(); ts.append((time.perf_counter() - t) * 1e3)
    print(name, " ".join(f"{t:.2f}" for t in ts), "ms")
# display loop: render + peek per frame, deferral 16
for defer in (16, 0):
    rt.ctx.set_option(abi.OPT_DEFER, defer)
    rt.ctx.synchronize()
    t = time.perf_counter()
    for _ in range(128):
        rt.render(); rt.ctx.peek_result()
    rt.ctx.synchronize()
    print(f"render+peek loop, defer {defer}: {(time.perf_counter() - t) * 1e3 / 128:.3f} ms per frame")
    t = time.perf_counter()
    for _ in range(128):
        rt.render(); rt.ctx.read_result()
    rt.ctx.synchronize()
    print(f"render+read_result loop, defer {defer}: {(time.perf_counter() - t) * 1e3 / 128:.3f} ms per frame")
