"""When do the waves of the bounce kernel leave? Needs a library built with -DOKRT_TAIL_PROBE
(apply tools/experiments/r05_drain_merge.diff, then tools/build_variant.sh tail okrt_fast_trace.hip -DOKRT_TAIL_PROBE -DOKRT_TRACE_MERGE=0; OKRT_LIBRARY=gpu-raytracer_amd/libokrt_tail.so).
One K-frame launch set of a workload: full frame and one rank's share of an 8-way split; histogram of wave exit times."""
import ctypes, importlib, os, sys, time
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
builder, kwargs, W, H, desc = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "sponza"]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
lib = ctypes.CDLL(os.environ["OKRT_LIBRARY"])
probe = getattr(lib, "okrt_debug_tail_probe", None)
import numpy as np
buf = np.zeros((2, 16384), np.uint64)
ptr = buf.ctypes.data_as(ctypes.c_void_p)
for name, tile, width in (("full frame", (H, 0, 1), 5000), ("rank 0 of 8", (bench.STRIPE_ROWS, 0, 8), 1000), ("rank 3 of 8", (bench.STRIPE_ROWS, 3, 8), 1000)):
    rt.ctx.set_tile(*tile)
    rt.resetAccumulation()
    for _ in range(40):
        rt.renderFrames(K)
    rt.ctx.synchronize()
    for rep in range(3):
        if probe: probe(ptr)
        t0 = time.perf_counter()
        rt.renderFrames(K)
        rt.ctx.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        if probe: probe(ptr)
        line = f"{name}: launch set {wall:.3f} ms wall"
        if probe:
            m = buf[1] > 0
            start, end = buf[0][m].astype(np.float64) / 100.0, buf[1][m].astype(np.float64) / 100.0
            t0_ = start.min()
            ex = np.sort(end - t0_)
            line += (f"; bounce kernel: {m.sum()} waves, last start {start.max() - t0_:.0f} us, exits: first {ex[0]:.0f}, 10 % {ex[len(ex) // 10]:.0f}, median {ex[len(ex) // 2]:.0f}, "
                     f"90 % {ex[len(ex) * 9 // 10]:.0f}, 99 % {ex[len(ex) * 99 // 100]:.0f}, last {ex[-1]:.0f} us")
        print(line)
        if rep == 2 and probe:
            h = np.histogram(ex, bins=np.arange(0, ex[-1] + width / 100.0, width / 100.0))[0]
            nz = np.nonzero(h)[0][0]
            print(f"   waves leaving per {width / 100:.0f}-us bin from {nz * width / 100:.0f} us:", h[nz:].tolist())
rt.close()
