"""Developer timing loop (not the graded bench): python tools/quick_bench.py sponza 1920 1080 [frames]"""
import importlib, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 16
modes = sys.argv[5].split(",") if len(sys.argv) > 5 else ["fast", "ref"]
t0 = time.time()
desc = {"sponza": lambda: scenes.sponza_proxy(), "sponza_small": lambda: scenes.sponza_proxy(tex_size=128, env_face=128, tri_scale=0.1),
        "dragon80k": lambda: scenes.dragon_proxy(80000), "dragon": lambda: scenes.dragon_proxy(871414),
        "glass": lambda: scenes.glass_revolver_proxy(), "spheres": scenes.test_scene_spheres}[name]()
print(f"scene {desc.name}: {desc.num_triangles} tris, {len(desc.meshes)} meshes, built in {time.time()-t0:.1f}s", flush=True)
t0 = time.time()
rt, sc, res = scenes.instantiate(desc, W, H)
print(f"host tables (BVH+octree+upload) in {time.time()-t0:.1f}s", flush=True)
for mode in modes:
    rt.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_FAST if mode == "fast" else abi.TRAVERSAL_REFERENCE)
    rt.resetAccumulation()
    rt.ctx.set_option(abi.OPT_COUNTERS, 1); rt.ctx.reset_counters()
    t0 = time.time(); rt.render(); rt.ctx.synchronize()
    c = rt.ctx.counters()
    print(f"[{mode}] first frame (incl. structure build) {time.time()-t0:.2f}s counters {c}", flush=True)
    rt.ctx.set_option(abi.OPT_COUNTERS, 0)
    n = frames if mode == "fast" else max(1, frames // 8)
    for _ in range(2): rt.render()
    rt.ctx.timer_begin()
    for _ in range(n): rt.render()
    rt.ctx.timer_end(); ms = rt.ctx.timer_elapsed_ms() / n
    print(f"[{mode}] {ms:.3f} ms/frame  {c['segments']/ms/1e3:.1f} Mrays/s", flush=True)
acc = rt.ctx.read_result()
from PIL import Image
os.makedirs("gpurun_out", exist_ok=True)
Image.fromarray(acc[..., :3]).save(f"gpurun_out/{name}.png")
