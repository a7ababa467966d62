"""Per-kernel time per frame against the launch-set size F (okrt_render_frames): where the single-frame cost sits.
usage: python tools/frame_scaling.py [workload] [F ...]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "sponza"
fs = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8, 16, 32, 64]
builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
rt.ctx.set_option(abi.OPT_DEFER, 0)
rt.renderFrames(64)
print(f"{wl} {W}x{H}: ms per frame by kernel interval (camera, -, trace, shade+accumulate)")
for F in fs:
    n = max(128 // F, 2)
    rt.renderFrames(F)
    rt.ctx.synchronize()
    rt.ctx.set_option(abi.OPT_TIMING, n)
    rt.ctx.timer_begin()
    for _ in range(n): rt.renderFrames(F)
    rt.ctx.timer_end()
    ms, launches, frames = rt.ctx.interval_ms()
    tot = rt.ctx.timer_elapsed_ms()
    rt.ctx.set_option(abi.OPT_TIMING, 0)
    print(f"F={F:3d}: " + " ".join(f"{m / frames:8.4f}" for m in ms) + f"   sum {sum(ms) / frames:7.4f}  stream {tot / frames:7.4f} ms/frame ({launches} sets)", flush=True)
