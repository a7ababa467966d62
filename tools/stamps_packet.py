"""Diagnostic: where do the trace kernels' wave-cycles go? (needs libokrt_stamps.so: tools/build_variant.sh stamps okrt_fast_trace.hip -DOKRT_STAMPS)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OKRT_LIBRARY"] = "gpu-raytracer_amd/libokrt_stamps.so"
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.sponza_proxy()
rt, sc, res = scenes.instantiate(desc, 1920, 1080)
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for mb in ((0, 1) if os.environ.get('OKRT_STAMPS_PACKET') else (1,)):
    rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb)
    rt.resetAccumulation(); rt.render(); rt.render()
    rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.ctx.reset_counters(); rt.renderFrames(FRAMES); c = rt.ctx.counters(); rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    if mb == 0:
        tot = c["n_tri"]
        print(f"packet kernel: wave-cycles {tot:.3e}: setup {c['n_oct']/tot:.3f} node {c['n_mesh']/tot:.3f} leaf+pop {c['n_node']/tot:.3f} | tiles {c['mesh_hits']} "
              f"node steps/tile {c['sphere_hits']/max(c['mesh_hits'],1):.1f} ({c['n_mesh']/max(c['sphere_hits'],1):.0f} cycles each) leaf tris/tile {c['tex_taps']/max(c['mesh_hits'],1):.1f} "
              f"({c['n_node']/max(c['tex_taps'],1):.0f} cycles each incl. pops) setup {c['n_oct']/max(c['mesh_hits'],1):.0f} cycles/tile")
        continue
    tot = c["n_tri"]
    print(f"max_bounces={mb} segments {c['segments']} wave-cycles total {tot:.3e}: refill {c['n_oct']/tot:.3f} node loop {c['n_mesh']/tot:.3f} leaf {c['n_node']/tot:.3f} | "
          f"node loop: {c['mesh_hits']} wave-iterations, {c['sphere_hits']/max(c['mesh_hits'],1):.1f} active lanes each, {c['n_mesh']/max(c['mesh_hits'],1):.0f} cycles each | "
          f"leaf loop: {c['tex_taps']} wave-iterations, {c['env_taps']/max(c['tex_taps'],1):.1f} active lanes each, {c['n_node']/max(c['tex_taps'],1):.0f} cycles each | "
          f"nodes/ray {c['fast_nodes']/c['segments']:.1f} tris/ray {c['fast_tris']/c['segments']:.1f}")
    if os.environ.get("OKRT_STAMPS_ITERS"):
        print(f"   outer iterations {c['n_node']}, live lanes per iteration {c['n_oct']/max(c['n_node'],1):.1f}")
