"""Diagnostic: how full do the two step blocks of the bounce-ray kernel run? (needs libokrt_stamps.so:
tools/build_variant.sh stamps okrt_fast_trace.hip -DOKRT_STAMPS_BLOCKS; OKRT_FAST_DUAL=0|1 selects the kernel)"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
os.environ["OKRT_LIBRARY"] = "gpu-raytracer_amd/libokrt_stamps.so"
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
for wl in (sys.argv[1:] or ["sponza"]):
    builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
    rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
    ctx = rt.ctx
    os.environ.setdefault("OKRT_FAST_PACKET", "1")
    ctx.set_option(abi.OPT_COUNTERS, 1)
    def run(mb):
        ctx.set_option(abi.OPT_MAX_BOUNCES, mb); rt.resetAccumulation(); ctx.reset_counters(); rt.renderFrames(16); return ctx.counters()
    run(1)
    c1, c0 = run(1), run(0)
    d = {k: c1[k] - c0[k] for k in c1}  # the camera segment's own stamps drop out
    it = d["segments"] >> 32
    rays = d["segments"] & 0xFFFFFFFF
    nw, nl, tw, tl = d["n_oct"], d["n_mesh"], d["n_node"], d["n_tri"]
    live = d["pixels"]
    print(f"{wl}: bounce rays {rays}: node block {nw} wave-executions x {nl / max(nw, 1):.1f} lanes; triangle block {tw} x {tl / max(tw, 1):.1f} lanes; "
          f"iterations {it} (blocks per iteration {(nw + tw) / max(it, 1):.2f}, live lanes {live / max(it, 1):.1f}); lane-level steps per ray: node {nl / rays:.2f} tri {tl / rays:.2f}; "
          f"cost-weighted utilisation {(176 * nl + 95 * tl) / (64 * (176 * nw + 95 * tw)):.3f}", flush=True)
    rt.close()
