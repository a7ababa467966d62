"""Per-kernel time of ONE launch set on a rank's tile against the same kernels' share of the full frame: where a small tile's
fixed cost sits (camera kernel tail / bounce kernel tail / the rest).  usage: python tools/tile_kernels.py [ranks] [frames ...]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames_list = [int(a) for a in sys.argv[2:]] or [10, 20, 64]
builder, kwargs, W, H, _ = bench.WORKLOADS["sponza"]
rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
ctx = rt.ctx
ctx.set_option(abi.OPT_PIPELINE, 0)  # one set at a time: the kernels' own durations
def kernels(F):
    rt.renderFrames(F); rt.renderFrames(F); ctx.synchronize()
    ctx.set_option(abi.OPT_TIMING, 8)
    for _ in range(8): rt.renderFrames(F)
    ms, launches, frames = ctx.interval_ms()
    ctx.set_option(abi.OPT_TIMING, 0)
    return [m / launches for m in ms]
for F in frames_list:
    ctx.set_tile(H, 0, 1); full = kernels(F)
    ctx.set_tile(bench.STRIPE_ROWS, 3, ranks); tile = kernels(F)
    share = ctx.tile_rows() / H
    names = ["camera", "(camera shade)", "bounce trace", "shade + accumulate"]
    print(f"F={F}: tile = {share:.4f} of the frame")
    for n, f, t in zip(names, full, tile):
        print(f"   {n:20s} full {f:8.4f} ms  x share {f * share:7.4f}  tile {t:7.4f}  fixed cost {t - f * share:+.4f} ms")
    print(f"   total: full {sum(full):.4f}  tile {sum(tile):.4f}  ideal {sum(full) * share:.4f}  overhead {sum(tile) - sum(full) * share:+.4f} ms")
rt.close()
