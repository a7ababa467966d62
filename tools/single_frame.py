import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.sponza_proxy()
rt, sc, res = scenes.instantiate(desc, 1920, 1080)
rt.ctx.set_option(abi.OPT_DEFER, 0)
for _ in range(8): rt.render()
rt.ctx.synchronize()
rt.ctx.set_option(abi.OPT_TIMING, 64)
t0 = time.perf_counter()
for _ in range(64): rt.render()
rt.ctx.synchronize()
wall = (time.perf_counter() - t0) / 64 * 1e3
tr, sh, ot, n = rt.ctx.stage_ms()
print(f"single-frame launches: wall {wall:.4f} ms/frame; kernels: trace {tr/n:.4f} shade {sh/n:.4f} other {ot/n:.4f} sum {(tr+sh+ot)/n:.4f} ({n} frames)")
