"""Builder comparison (developer tool): python tools/build_bench.py [sponza|dragon|dragon80k|glass] [W H]
For each builder: build time (device timeline / host wall clock), structure size, traversal cost and ms/frame."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime: torch's must be loaded first)
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
desc = {"sponza": lambda: scenes.sponza_proxy(), "dragon80k": lambda: scenes.dragon_proxy(80000), "dragon": lambda: scenes.dragon_proxy(871414),
        "glass": lambda: scenes.glass_revolver_proxy()}[name]()
rt, sc, res = scenes.instantiate(desc, W, H)
for builder, label in ((abi.BUILDER_HOST, "host SAH"), (abi.BUILDER_DEVICE, "device PLOC"), (abi.BUILDER_DEVICE, "device PLOC (2nd)")):
    rt.ctx.set_option(abi.OPT_BUILDER, 1 - builder); rt.ctx.set_option(abi.OPT_BUILDER, builder)  # force a rebuild
    t0 = time.time(); rt.ctx.commit(); wall = (time.time() - t0) * 1e3
    bi = rt.ctx.build_info()
    rt.resetAccumulation()
    rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.ctx.reset_counters(); rt.render(); c = rt.ctx.counters(); rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    rt.renderFrames(64)
    rt.ctx.timer_begin(); rt.renderFrames(64); rt.ctx.timer_end(); ms = rt.ctx.timer_elapsed_ms() / 64
    print(f"{name} {label:18s} build {bi['device_ms']:8.2f} ms device / {wall:8.2f} ms wall  nodes {bi['num_nodes']:7d} tris {bi['num_triangles']:7d} depth {bi['max_depth']:3d}"
          f"  visits/ray {c['fast_nodes']/c['segments']:.1f}+{c['fast_tris']/c['segments']:.1f}  {ms:.3f} ms/frame", flush=True)
