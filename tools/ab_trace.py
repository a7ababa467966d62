"""A/B of the traversal kernels: per-kernel ms per frame at launch sets of 64 and 1 frames, plus node / triangle steps per bounce ray.
usage: [OKRT_FAST_DUAL=0|1 OKRT_LIBRARY=...] python tools/ab_trace.py [workload ...]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
for wl in (sys.argv[1:] or ["sponza"]):
    builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
    rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
    ctx = rt.ctx
    ctx.set_option(abi.OPT_DEFER, 0)
    def counted(mb):
        ctx.set_option(abi.OPT_MAX_BOUNCES, mb); ctx.set_option(abi.OPT_COUNTERS, 1); rt.resetAccumulation(); ctx.reset_counters(); rt.render()
        c = ctx.counters(); ctx.set_option(abi.OPT_COUNTERS, 0); ctx.set_option(abi.OPT_MAX_BOUNCES, 1); return c
    full, cam = counted(1), counted(0)
    br = max(full["segments"] - full["pixels"], 1)
    tiles = max(cam["pixels"] / 64, 1)
    bi = ctx.build_info()
    line = (f"{wl}: {bi['num_nodes']} nodes depth {bi['max_depth']}, camera tile: {cam['fast_nodes'] / 64 / tiles:.2f} node steps {cam['fast_tris'] / 64 / tiles:.2f} tris; "
            f"bounce rays {br}, nodes/ray {(full['fast_nodes'] - cam['fast_nodes']) / br:.2f}, tris/ray {(full['fast_tris'] - cam['fast_tris']) / br:.2f} |")
    rt.resetAccumulation()
    rt.renderFrames(64)
    for F in (64, 1):
        n = max(256 // F, 4) if F > 1 else 32
        rt.renderFrames(F); ctx.synchronize()
        ctx.set_option(abi.OPT_TIMING, n)
        ctx.timer_begin()
        for _ in range(n): rt.renderFrames(F)
        ctx.timer_end()
        ms, launches, frames = ctx.interval_ms()
        tot = ctx.timer_elapsed_ms()
        ctx.set_option(abi.OPT_TIMING, 0)
        line += f" F={F}: cam {ms[0] / frames:.4f} trace {ms[2] / frames:.4f} shade {ms[3] / frames:.4f} total {tot / frames:.4f} |"
    print(line, flush=True)
    rt.close()
