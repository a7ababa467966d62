"""One-off fuzz: the reference-order kernel against the oracle under random uniform / non-uniform instance scale in (0.3, 2.5) (quirk Q3): images and the reference traversal counters. usage: python tools/fuzz_scale.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import conftest
from conftest import abi, scenes, compare_accum
oracle = conftest.Oracle()
rng = np.random.default_rng(5)
worst = (1.0, None)
for c in range(150):
    sc = [tuple(float(x) for x in rng.uniform(0.3, 2.5, 3)) if rng.integers(0, 2) else float(rng.uniform(0.3, 2.5)) for _ in range(3)]
    kw = dict(seed=int(rng.integers(1, 10**6)), n_objects=int(rng.integers(1, 7)), detail=int(rng.integers(3, 12)), transforms=True, with_lights=bool(rng.integers(0, 2)), octree=(0, 50), scales=sc)
    W, H, mb = int(rng.integers(17, 160)), int(rng.integers(9, 100)), int(rng.integers(0, 4))
    desc = scenes.small_mesh_scene(**kw)
    ort, _, _ = oracle.instantiate(desc, W, H); ort.ctx.set_option(abi.OPT_MAX_BOUNCES, mb); ort.ctx.set_option(abi.OPT_COUNTERS, 1); ort.render(); ca = ort.ctx.read_accum(); cc = ort.ctx.counters(); ort.close()
    rt, _, _ = scenes.instantiate(desc, W, H); rt.ctx.set_option(abi.OPT_MAX_BOUNCES, mb); rt.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_REFERENCE); rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.render(); ga = rt.ctx.read_accum(); gc = rt.ctx.counters(); rt.close()
    frac, rmse, _, _ = compare_accum(ga, ca)
    if frac < worst[0]: worst = (frac, (c, kw, W, H, mb))
    bad = frac < 0.985 or any(abs(gc[k] - cc[k]) > 0.01 * max(cc[k], 1) + 16 for k in ("n_node", "n_tri", "n_mesh", "mesh_hits"))
    if bad: print("SUSPECT", frac, rmse, {k: (gc[k], cc[k]) for k in ("n_node", "n_tri", "mesh_hits")}, kw, W, H, mb, flush=True)
print("worst", worst)
