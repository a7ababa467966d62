"""How the camera-packet traversal scales with resident waves (is it latency- or issue-bound?): the UNFUSED camera segment
(OKRT_FAST_FUSED=0: generate + tracePacketKernel + shadeKernel<FIRST>) timed per kernel interval.
usage: OKRT_FAST_FUSED=0 [OKRT_FAST_PACKET_BLOCKS=n OKRT_LIBRARY=...libokrt_pw8.so] python tools/packet_occupancy.py [workload ...]
(libokrt_pw8.so: tools/build_variant.sh pw8 okrt_fast_trace.hip -DOKRT_PACKET_WAVES=8; round 4: 5 -> 8 waves per SIMD = -4.5 %: the kernel is issue-bound, VALU + scalar unit)"""
import importlib, os, sys
sys.path.insert(0, os.getcwd())
abi = importlib.import_module("gpu-raytracer_amd._abi"); scenes = importlib.import_module("gpu-raytracer_amd.scenes")
import bench
for wl in (sys.argv[1:] or ["sponza"]):
    builder, kwargs, W, H, _ = bench.WORKLOADS[wl]
    rt, sc, res = scenes.instantiate(getattr(scenes, builder)(**kwargs), W, H)
    ctx = rt.ctx
    ctx.set_option(abi.OPT_DEFER, 0)
    F, n = 16, 8
    rt.renderFrames(F); rt.renderFrames(F); ctx.synchronize()
    ctx.set_option(abi.OPT_TIMING, n)
    for _ in range(n): rt.renderFrames(F)
    ms, launches, frames = ctx.interval_ms()
    ctx.set_option(abi.OPT_TIMING, 0)
    print(f"{wl} lib={os.path.basename(os.environ.get('OKRT_LIBRARY', 'libokrt.so'))} fused={os.environ.get('OKRT_FAST_FUSED', '1')} "
          f"blocks={os.environ.get('OKRT_FAST_PACKET_BLOCKS', '6')}: camera trace {ms[0] / frames:.4f} camera shade {ms[1] / frames:.4f} bounce trace {ms[2] / frames:.4f} shade {ms[3] / frames:.4f} ms/frame", flush=True)
    rt.close()
