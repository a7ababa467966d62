"""Feasibility probe for the frame ring: R contexts (R HIP streams) of one process each render whole frames in launch
sets of G, round-robin; aggregate ms per frame. Emulates R slots in flight without the in-order accumulation chain.
usage: python tools/ring_probe.py [workload]"""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "sponza"
builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
desc = getattr(scenes, builder)(**kwargs)
ctxs = []
for r in range(4):
    rt, sc, res = scenes.instantiate(desc, W, H)
    rt.ctx.set_option(abi.OPT_DEFER, 0)
    rt.renderFrames(16); rt.ctx.synchronize()
    ctxs.append((rt, sc, res))
for R in (1, 2, 3, 4):
    for G in (1, 2, 4, 8, 16):
        total = 256
        sets = total // G
        for k in range(R): ctxs[k][0].renderFrames(G)
        for k in range(R): ctxs[k][0].ctx.synchronize()
        t0 = time.perf_counter()
        for i in range(sets): ctxs[i % R][0].renderFrames(G)
        for k in range(R): ctxs[k][0].ctx.synchronize()
        print(f"{wl} R={R} G={G:2d}: {(time.perf_counter() - t0) * 1e3 / (sets * G):.4f} ms/frame", flush=True)
