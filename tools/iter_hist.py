"""Diagnostic: distribution of traversal steps per bounce ray (traceKernel) and per camera tile (packet kernel).
Needs libokrt_hist.so: tools/build_variant.sh hist okrt_fast_trace.hip -DOKRT_ITER_HIST.  usage: python tools/iter_hist.py [workload]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
os.environ["OKRT_LIBRARY"] = "gpu-raytracer_amd/libokrt_hist.so"
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "sponza"
builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
names = ["<=16", "<=32", "<=64", "<=128", "<=256", "<=512", "<=1024", ">1024"]
keys = ["n_oct", "n_mesh", "n_node", "n_tri", "mesh_hits", "sphere_hits", "tex_taps", "env_taps"]
for mb, what in ((0, "camera tiles (packet kernel): node steps + triangle tests per tile"), (1, "camera tiles + bounce rays (traceKernel): steps per ray")):
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
    rt.resetAccumulation(); rt.render()
    rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.ctx.reset_counters(); rt.resetAccumulation(); rt.render(); c = rt.ctx.counters(); rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    print(f"{wl} max_bounces={mb}: {what}; max {c['fast_nodes']}; stack entries spilled to the HBM slab by the per-ray kernel: {c['fast_tris']} "
          f"({c['fast_tris'] * 8 / 1e6:.1f} MB of 8-byte writes per frame)")
    print("   " + "  ".join(f"{n}: {c[k]}" for n, k in zip(names, keys)))
