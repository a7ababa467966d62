"""One okrt_render per frame (the reference's main loop): ms per frame at deferral 0 ... 16 and launch sets of 32, with launch
sets in flight together (OKRT_OPT_PIPELINE; OKRT_PIPELINE_DEPTH=2..4 in the environment picks how many) and without.
usage: python tools/percall_probe.py [workload ...]"""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
for wl in (sys.argv[1:] or ["sponza"]):
    builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
    rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
    ctx = rt.ctx
    rt.renderFrames(64); rt.renderFrames(16); rt.renderFrames(16)
    out = []
    for pipe in (0, 32):
        ctx.set_option(abi.OPT_PIPELINE, pipe)
        for defer in (0, 2, 4, 8, 16):
            ctx.set_option(abi.OPT_DEFER, defer)
            for _ in range(32): rt.render()
            ctx.synchronize()
            t = time.perf_counter()
            for _ in range(128): rt.render()
            ctx.synchronize()
            out.append(f"p{pipe} d{defer}: {(time.perf_counter() - t) * 1e3 / 128:.4f}")
        for F in (32,):
            rt.renderFrames(F); ctx.synchronize()
            t = time.perf_counter()
            for _ in range(256 // F): rt.renderFrames(F)
            ctx.synchronize()
            out.append(f"p{pipe} sets of {F}: {(time.perf_counter() - t) * 1e3 / 256:.4f}")
    print(wl, "depth", os.environ.get("OKRT_PIPELINE_DEPTH", "4"), "ms per frame |", " | ".join(out), flush=True)
    rt.close()
