import importlib, sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
desc = scenes.small_mesh_scene(with_lights=False)
W, H = 160, 90
for mb in (0, 1):
    accs = []
    for trav in (abi.TRAVERSAL_REFERENCE, abi.TRAVERSAL_FAST):
        rt, _, _ = scenes.instantiate(desc, W, H)
        rt.ctx.set_option(abi.OPT_TRAVERSAL, trav); rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb); rt.ctx.set_option(abi.OPT_COUNTERS, 1)
        rt.render()
        accs.append(rt.ctx.read_accum()); print(mb, trav, rt.ctx.counters())
        rt.close()
    same = (accs[0] == accs[1]).all(-1)
    print("max_bounces", mb, "identical pixels", same.mean())
    ys, xs = np.nonzero(~same)
    for y, x in list(zip(ys, xs))[:6]:
        print("  px", x, y, accs[0][y, x], accs[1][y, x])
    rows = (~same).sum(1); print("  diff per row (first 30):", rows[:30])
