"""Diagnostic: where do the trace kernels' wave-cycles go? (needs libokrt_stamps.so built with -DOKRT_STAMPS)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OKRT_LIBRARY"] = "gpu-raytracer_amd/libokrt_stamps.so"
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.sponza_proxy()
rt, sc, res = scenes.instantiate(desc, 1920, 1080)
for mb in (0, 1):
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
    rt.resetAccumulation(); rt.render(); rt.render()
    rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.ctx.reset_counters(); rt.render(); c = rt.ctx.counters(); rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    tot = c["n_tri"]
    print(f"max_bounces={mb} segments {c['segments']} waves-cycles total {tot:.3e}: refill {c['n_oct']/tot:.3f} inner {c['n_mesh']/tot:.3f} leaf {c['n_node']/tot:.3f} "
          f"| refills {c['mesh_hits']} iters {c['sphere_hits']} avg active lanes/iter {c['tex_taps']/max(c['sphere_hits'],1):.1f} "
          f"| cycles/refill {c['n_oct']/max(c['mesh_hits'],1):.0f} cycles/iter {(c['n_mesh']+c['n_node'])/max(c['sphere_hits'],1):.0f} nodes/ray {c['fast_nodes']/c['segments']:.1f} tris/ray {c['fast_tris']/c['segments']:.1f}")
