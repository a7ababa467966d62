"""Can the tiles that emit the very long bounce rays be known from the previous launch set?  Region of the sponza frame, two
consecutive sets of F frames: nodes visited per segment-1 ray (okrt_debug_kat through the fast structure), per 8x8 tile the maximum
over the set.  Printed: how many of set B's longest rays left tiles that set A would have flagged, for several flag sizes.
usage: python tools/experiments/r05_long_ray_history.py [F=8] [x0 y0 w h]"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
F = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x0, y0, w, h = (int(a) for a in sys.argv[2:6]) if len(sys.argv) >= 6 else (0, 0, 1920, 1080)
rt, sc, res = scenes.instantiate(scenes.sponza_proxy(), 1920, 1080)
rt.render(); rt.ctx.synchronize()
bits = lambda a: np.ascontiguousarray(a, dtype=np.uint32).view(np.float32)
h8, w8 = h // 8 * 8, w // 8 * 8
ys, xs = np.mgrid[y0:y0 + h8, x0:x0 + w8]
px, py = xs.ravel().astype(np.uint32), ys.ravel().astype(np.uint32)
def visited(frame):
    rd = rt.render_data
    rd.numAccumulationFrames = frame
    rdf = np.frombuffer(bytes(rd), np.float32)
    n = len(px)
    out = np.zeros(n)
    step = 1 << 19
    for a in range(0, n, step):
        b = min(n, a + step)
        rec = np.concatenate([bits(px[a:b])[:, None], bits(py[a:b])[:, None], bits(np.full(b - a, 1, np.uint32))[:, None], np.broadcast_to(rdf, (b - a, 60))], 1)
        o = rt.ctx.kat(abi.KAT_SEGMENT_RAY, rec, 8)
        alive = o[:, 0].view(np.uint32) == 1
        recs = np.concatenate([o[:, 1:4], o[:, 4:7], bits(np.zeros(b - a, np.uint32))[:, None]], 1).astype(np.float32)
        r = rt.ctx.kat(abi.KAT_CLOSEST_HIT_FAST, recs, 5)
        v = r[:, 2].view(np.uint32).astype(np.float64) + r[:, 3].view(np.uint32).astype(np.float64) * 0  # nodes visited
        v[~alive] = 0
        out[a:b] = v
    return out.reshape(h8, w8)
tmax = lambda v: v.reshape(h8 // 8, 8, w8 // 8, 8).max(3).max(1)
A = np.zeros((h8 // 8, w8 // 8)); Bmax = np.zeros_like(A); Brays = []
for f in range(F):
    A = np.maximum(A, tmax(visited(3 + f)))
for f in range(F):
    v = visited(3 + F + f)
    Bmax = np.maximum(Bmax, tmax(v))
    Brays.append(v)
B = np.stack(Brays)
allv = B[B > 0]
print(f"{F}-frame sets, {A.size} tiles; set B: {allv.size} rays, mean {allv.mean():.1f} nodes, 99 % {np.percentile(allv, 99):.0f}, 99.9 % {np.percentile(allv, 99.9):.0f}, 99.99 % {np.percentile(allv, 99.99):.0f}, max {allv.max():.0f}")
print(f"per-tile set maximum, A vs B: correlation {np.corrcoef(A.ravel(), Bmax.ravel())[0, 1]:.3f}")
order = np.argsort(-A.ravel())
for frac in (0.05, 0.1, 0.2, 0.3, 0.5):
    flagged = np.zeros(A.size, bool); flagged[order[: int(frac * A.size)]] = True
    fl = flagged.reshape(A.shape)
    fl_px = np.repeat(np.repeat(fl, 8, 0), 8, 1)
    line = f"flag the top {100 * frac:.0f} % of the tiles by set A's maximum (>= {A.ravel()[order[int(frac * A.size) - 1]]:.0f} nodes): "
    for q in (99, 99.9, 99.99):
        thr = np.percentile(allv, q)
        long_ = B >= thr
        line += f"{100 * (long_ & fl_px[None]).sum() / long_.sum():.0f} % of B's rays >= {thr:.0f} nodes; "
    unfl = B[:, ~fl_px]
    line += f"longest ray from an unflagged tile {unfl.max():.0f} (overall {B.max():.0f})"
    print(line)
rt.close()
