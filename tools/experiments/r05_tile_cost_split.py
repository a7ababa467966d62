"""Which camera tiles would be cheaper ray by ray?  Per 8x8 tile of the sponza 1080p frame: what the packet kernel recorded (node steps +
triangle tests, one wave-instruction stream for 64 rays) against what its 64 rays cost one by one (okrt_debug_kat through the fast structure:
nodes visited + triangles tested per ray), priced with the per-ray kernel's measured lane use.
Needs a build that dumps the recorded costs: apply tools/experiments/r05_tile_cost_dump.diff, tools/build_variant.sh tilecost okrt_fast_pipeline.hip -DOKRT_DEBUG_TILECOST,
OKRT_LIBRARY=gpu-raytracer_amd/libokrt_tilecost.so OKRT_FAST_DUMP_TILECOST=/tmp/tilecost.bin python tools/experiments/r05_tile_cost_split.py"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "sponza"
builder, kwargs, W, H, desc = bench.WORKLOADS[wl]
rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
rt.ctx.set_option(abi.OPT_MAX_BOUNCES, 0)
for _ in range(3):
    rt.renderFrames(4)
rt.ctx.synchronize()
cost = np.fromfile(os.environ["OKRT_FAST_DUMP_TILECOST"], np.uint32).astype(np.float64)
tw, th = (W + 7) // 8, (H + 7) // 8
print(f"{wl}: {cost.size} tiles ({tw} x {th}); packet steps per tile: mean {cost.mean():.1f}, median {np.median(cost):.0f}, 90 % {np.percentile(cost, 90):.0f}, 99 % {np.percentile(cost, 99):.0f}, max {cost.max():.0f}")
# per-ray cost of the same camera rays
bits = lambda a: np.ascontiguousarray(a, dtype=np.uint32).view(np.float32)
H8, W8 = th * 8, tw * 8
ys, xs = np.mgrid[0:H8, 0:W8]
inside = (ys < H) & (xs < W)
px, py = np.minimum(xs, W - 1).ravel().astype(np.uint32), np.minimum(ys, H - 1).ravel().astype(np.uint32)
rd = rt.render_data
rd.numAccumulationFrames = 3
rdf = np.frombuffer(bytes(rd), np.float32)
n = len(px)
nodes = np.zeros(n); tris = np.zeros(n)
step = 1 << 19
for a in range(0, n, step):
    b = min(n, a + step)
    rec = np.concatenate([bits(px[a:b])[:, None], bits(py[a:b])[:, None], bits(np.zeros(b - a, np.uint32))[:, None], np.broadcast_to(rdf, (b - a, 60))], 1)
    o = rt.ctx.kat(abi.KAT_SEGMENT_RAY, rec, 8)
    recs = np.concatenate([o[:, 1:4], o[:, 4:7], bits(np.zeros(b - a, np.uint32))[:, None]], 1).astype(np.float32)
    r = rt.ctx.kat(abi.KAT_CLOSEST_HIT_FAST, recs, 5)
    nodes[a:b] = r[:, 2].view(np.uint32)  # (the probe reports nodes only: triangle tests are priced as a share of the node work below)
nodes = np.where(inside.ravel(), nodes, 0).reshape(th, 8, tw, 8)
# the tile order of the cost array: row-major tiles of the frame
rn = nodes.sum((1, 3)).ravel()
tau = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2  # triangle-test work per unit of node work, per-ray kernel (sponza ~0.2, dragon ~0.65: bench counters)
# price: a packet step ~ 130 VALU wave-instructions per node step, 95 per triangle test (recorded as one sum: priced at the mix of the frame);
# per ray: 64 lanes share a wave-instruction at the per-ray kernel's lane use (0.57)
for util in (0.57, 0.7):
    per_ray = rn * 130 * (1 + tau) / 64 / util
    packet = cost * 120
    better = per_ray < packet
    print(f"lane use {util}: per-ray cheaper on {100 * better.mean():.1f} % of the tiles; packet work {packet.sum():.3e}, of which in those tiles {100 * packet[better].sum() / packet.sum():.1f} %; "
          f"tracing them per ray instead: {per_ray[better].sum():.3e} -> the camera traversal's work x {(packet[~better].sum() + per_ray[better].sum()) / packet.sum():.3f}")
    for thr in (2.0, 3.0):
        b2 = per_ray * thr < packet
        print(f"   only where per-ray is {thr}x cheaper: {100 * b2.mean():.1f} % of the tiles, {100 * packet[b2].sum() / packet.sum():.1f} % of the packet work -> x {(packet[~b2].sum() + per_ray[b2].sum()) / packet.sum():.3f}")
print(f"per tile: rays' nodes {rn.mean():.0f} (sum over 64 rays); correlation of packet steps with it {np.corrcoef(cost, rn)[0, 1]:.3f}")
rt.close()
