#!/usr/bin/env python3
"""bench.py — Mrays/s + ms/frame of the path-tracing hot path on N MI355X (one process per GPU).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sponza|sponza4k|dragon|dragon80k|glass|spheres]

A "step" is one frame = one sample per pixel of the whole frame. N = 1: the workload is
BASELINE.json's headline config (Sponza-class 1920x1080, configs[3]; procedural stand-in of
262 267 triangles / 390 meshes / 26 textures). N > 1: the SAME frame is split into
interleaved 8-row stripes, rank r owns stripes r, r+N, ...; scene tables are replicated; there is no
data-path collective per frame — the per-rank float4 accumulation tiles are gathered to rank 0 over
RCCL (okrt_mgpu_gather_accum[_all], behind the C ABI) every 64 frames and after the last one, inside
the timed region. Total work is fixed, so "scaling" is "strong".

N > 1 starts either way (gpu-raytracer_amd/mgpu.py: launch_plan), without torch:
  python bench.py --gpus N                                   one process drives N contexts (ncclCommInitAll, one group)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
                                                             one process per rank (RANK / WORLD_SIZE from the launcher);
                                                             the RCCL id travels through a file store
With fewer GPUs than ranks either form runs as a clearly marked REHEARSAL (ranks share GPUs).

The JSON line carries the roofline of the dominant kernel family (BVH traversal) and a CPU baseline
(the scalar oracle on this box's host cores, bounded sample, rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md:36)
# 128-byte L1 -> L2 read requests per second a pure random gather of 64-byte records sustains at the traversal kernel's occupancy
# (tools/calib/l1_gather_rate.hip -> profiles/r03_l1_gather_rate.txt): from a table that sits in L2 (2-4 MiB: 13.9 TB/s of records), and
# from one that sits in the Infinity Cache (64 MiB: 3.76 TB/s at 6 % L2 hits -> 56 G/s for the requests that miss L2). A kernel whose
# requests hit L2 with probability h cannot gather faster than 1 / (h / hit_rate + (1 - h) / miss_rate).
GATHER_L2_HIT_REQUESTS_PER_S = 215e9
GATHER_L2_MISS_REQUESTS_PER_S = 56e9


def gather_ceiling(l2_hit_fraction):
    h = min(max(l2_hit_fraction, 0.0), 1.0)
    return 1.0 / (h / GATHER_L2_HIT_REQUESTS_PER_S + (1.0 - h) / GATHER_L2_MISS_REQUESTS_PER_S)


L2_PEAK_GBS = 34500.0  # aggregate L2 bandwidth of the 8 XCDs (same guide, "L2 (per XCD)")
STRIPE_ROWS = 8
GATHER_EVERY = 64

WORKLOADS = {
    # name: (scene builder kwargs, width, height, description)
    "sponza": ("sponza_proxy", {}, 1920, 1080, "sponza-proxy 262267 tris/390 meshes/26 tex, 1920x1080, 1 spp/frame, NUM_BOUNCES=1"),
    "sponza4k": ("sponza_proxy", {}, 3840, 2160, "sponza-proxy 262267 tris/390 meshes/26 tex, 3840x2160, 1 spp/frame, NUM_BOUNCES=1"),
    "dragon": ("dragon_proxy", {"num_triangles": 871414}, 1280, 720, "dragon-proxy 871414 tris, diffuse, 1280x720"),
    # beyond the 256 MiB Infinity Cache: 384 MB of Tri48 + 40 MB of Node4Q (+ 512 MB of shading records): the one workload whose
    # fabric-side bytes are HBM bytes
    "dragon8m": ("dragon_proxy", {"num_triangles": 8000000}, 1280, 720, "dragon-proxy 8000000 tris, diffuse, 1280x720 (structure beyond the Infinity Cache)"),
    "dragon80k": ("dragon_proxy", {"num_triangles": 80000}, 1280, 720, "dragon-proxy 80000 tris, diffuse, 1280x720"),
    "glass": ("glass_revolver_proxy", {}, 1920, 1080, "glass + revolver-proxy + IBL, 1920x1080"),
    "spheres": ("test_scene_spheres", {}, 256, 256, "TestScene 4 spheres 256x256"),
}


def algorithmic_bytes(c):
    """bytes the reference's own structures would move for these counters."""
    traversal = 48 * c["n_oct"] + 232 * c["n_mesh"] + 36 * c["n_node"] + 36 * c["n_tri"]
    shading = 132 * c["mesh_hits"] + 16 * c["tex_taps"] + 16 * c["env_taps"] + 36 * c["pixels"]
    return traversal, shading


def kernel_source_sha():
    """Fingerprint of everything the device library is compiled from: a committed PMC measurement (profiles/*_pmc_*.json)
    is only attached to a bench line produced by the same kernels."""
    import glob
    import hashlib
    h = hashlib.sha256()
    pkg = os.path.join(REPO, "gpu-raytracer_amd")
    for f in sorted(glob.glob(os.path.join(pkg, "csrc", "*")) + [os.path.join(REPO, "include", "okrt.h"), os.path.join(pkg, "build.py")]):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def load_pmc(workload, frames_per_launch):
    """Newest profiles/r*_pmc_<workload>_f<F>.json whose stamp matches these kernel sources and this launch-set size, else
    (None, reason). The file is written by tools/pmc_traffic.py from rocprofv3 --pmc passes of this same bench command."""
    import glob
    best, why = None, "no profiles/r*_pmc_%s_f*.json" % workload
    sha = kernel_source_sha()
    for f in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_%s_f*.json" % workload))):
        try:
            j = json.load(open(f))
        except Exception:
            continue
        if j.get("source_sha") != sha:
            why = f"{os.path.basename(f)} was measured on other kernel sources ({j.get('source_sha')} != {sha})"
            continue
        if j.get("frames_per_launch") != frames_per_launch:
            why = f"{os.path.basename(f)} was measured with launch sets of {j.get('frames_per_launch')} frames, this run uses {frames_per_launch}"
            continue
        best = (os.path.basename(f), j)
    return (best, None) if best else (None, why)


def usable_cpus():
    """Threads worth starting: the CPU count, cut down to the scheduler affinity and the cgroup quota (a one-GPU box
    shows 256 CPUs but grants its job a share of them)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(desc, width, height, frame_index_rd, target_seconds=15.0):
    """The scalar oracle (oracle/liboracle.so) on the host cores, on a bounded sample of the same frame:
    every `stride`-th row, stride chosen from a short calibration run to cost about `target_seconds`."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import conftest
    oracle = conftest.Oracle()
    rt, scene, res = oracle.instantiate(desc, width, height)
    rt.calculateProjectionData()
    rt.updateBuffers()
    rd = rt.render_data
    cores = usable_cpus()
    abi = conftest.abi

    def run(stride):
        rt.ctx.set_tile(1, 0, stride)
        rt.ctx.set_option(abi.OPT_COUNTERS, 1)
        rt.ctx.reset_counters()
        t0 = time.perf_counter()
        rt.ctx.render(rd)
        dt = time.perf_counter() - t0
        return rt.ctx.counters()["segments"], dt, rt.ctx.tile_rows()

    # one thread first ("report 1-thread and all-thread"): ~64 rows spread over the frame
    oracle.lib.okrt_oracle_set_threads(rt.ctx.h, 1)
    rays1, dt1, rows1 = run(max(height // 64, 1))
    oracle.lib.okrt_oracle_set_threads(rt.ctx.h, cores)
    rays, dt, rows = run(max(height // 16, 1))  # calibration: ~16 rows
    rate = rays / max(dt, 1e-6)
    full_rays = rays * (height / max(rows, 1))
    full_seconds = full_rays / rate
    if full_seconds > target_seconds:  # few cores: a subset of the rows of one frame
        stride = max(1, int(round(full_seconds / target_seconds)))
        rays, dt, rows = run(stride)
        what = f"every {stride}th row of one {width}x{height} frame ({rows} rows)"
    else:  # many cores: whole frames, consecutive frame indices, until ~target_seconds
        frames = max(1, int(target_seconds / full_seconds))
        rt.ctx.set_tile(height, 0, 1)
        rt.ctx.set_option(abi.OPT_COUNTERS, 1)
        rt.ctx.reset_counters()
        t0 = time.perf_counter()
        for _ in range(frames):
            rt.updateBuffers()  # numAccumulationFrames += 1, like RayTracer::render
            rt.ctx.render(rt.render_data)
        dt = time.perf_counter() - t0
        rays = rt.ctx.counters()["segments"]
        what = f"{frames} whole {width}x{height} frames"
    rt.close()
    return {"value": round(rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "value_1thread": round(rays1 / dt1 / 1e6, 5),
            "sample": f"oracle (scalar C++ restatement, reference traversal order): {what}, {rays} rays in {dt:.1f} s on {cores} threads; "
                      f"1 thread: {rows1} rows, {rays1} rays in {dt1:.1f} s"}


def roofline_report(args, world, samples_split, interval_ms, launches, frames, build, ref_c, fast_c, cam_c, ref_trav_bytes,
                    ref_shade_bytes, trace_ms_frame, shade_ms_frame, other_ms_frame, terminal=False):
    """The roofline object of the JSON line (DESIGN.md §4 "Roofline accounting").

    Per kernel of the fast pipeline, live: average launch duration from the HIP events the library records around every
    kernel on the stream it runs on (okrt_get_interval_ms), and the kernel's ALGORITHMIC bytes per launch = what its own
    algorithm has to move through HBM at least once (ray / hit / queue / framebuffer records at their real sizes, one
    shading record per hit, 16 B per texture tap, the traversal structure once per launch) — a lower bound of its HBM
    traffic, so algorithmic_frac <= 1. `traffic` is the MEASURED HBM-side bytes per launch of the same kernel (rocprofv3
    --pmc FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_traffic.py -> profiles/r*_pmc_<workload>.json) and only attached when that
    file was measured on these kernel sources and this launch-set size; `frac` = traffic / time / peak when it is,
    else the algorithmic fraction with traffic = null and traffic_stale = true. The headline kernel is the one with the
    largest share of the frame."""
    per_gpu = world if not samples_split else 1  # counters are whole-frame sums; a rank traces 1 / world of the pixels
    launches = max(launches, 1)
    frames_per_launch = max(frames // launches, 1)
    structure_bytes = 64 * build["num_nodes"] + 48 * build["num_triangles"]
    pixels, segs = fast_c["pixels"], fast_c["segments"]
    bounce_rays = max(segs - pixels, 0)          # rays of segments 1..: queue entries the camera kernel / shading kernels append
    cam_nodes, cam_tris = cam_c["fast_nodes"], cam_c["fast_tris"]            # lane-level counts (x64 per wave-uniform visit)
    b_nodes, b_tris = fast_c["fast_nodes"] - cam_nodes, fast_c["fast_tris"] - cam_tris
    cam_hits, cam_taps, cam_env = cam_c["mesh_hits"], cam_c["tex_taps"], cam_c["env_taps"]
    later_hits, later_taps, later_env = fast_c["mesh_hits"] - cam_hits, fast_c["tex_taps"] - cam_taps, fast_c["env_taps"] - cam_env
    slot = 16 if frames_per_launch > 1 else 36   # a finished path: 16-byte radiance slot (F > 1) or accumulation RMW + resolve
    mb = args.max_bounces
    kernels = []

    def add(name, ms_sum, alg_frame, l2_frame, once=0, stream_frame=0):
        if ms_sum <= 0:
            return
        launch_ms = ms_sum / launches
        alg = alg_frame / per_gpu * frames_per_launch + once
        l2 = l2_frame / per_gpu * frames_per_launch
        kernels.append({"kernel": name, "launch_ms": round(launch_ms, 4), "ms_per_frame": round(launch_ms / frames_per_launch, 5),
                        "algorithmic_bytes_per_launch": int(alg), "algorithmic_gbs": round(alg / (launch_ms * 1e-3) / 1e9, 1),
                        # the part of the reads that is a coalesced 16-B-per-lane stream (queue entries, hit records, radiance slots):
                        # the only pattern FETCH_SIZE under-counts (x2, profiles/r03_pmc_calibration.txt); tools/pmc_traffic.py adds it back
                        "stream_read_bytes_per_launch": int(stream_frame / per_gpu * frames_per_launch),
                        "algorithmic_frac": round(alg / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        # bytes the kernel's per-lane gathers ask of the vector L1s (64 B per node visit, 48 B per triangle test, ...): an UPPER
                        # bound of what reaches L2 — the L1s' 128-byte lines serve about two thirds of a traversal's records.
                        # l2_read_requests / l2_frac / gather_ceiling_frac below are MEASURED (TCP_TCC_READ_REQ, attached from the PMC file)
                        "l1_gather_bytes_per_launch": int(l2), "l2_read_requests_per_launch": None, "l2_frac": None, "gather_ceiling_frac": None})

    if args.traversal == "fast" and len(interval_ms) >= 2:
        # camera-segment kernel: generates, traces (wave-uniform node / triangle loads: once per WAVE), shades, appends queue 1
        cam_surv = bounce_rays if mb == 1 else (cam_hits if mb > 1 else 0)  # (mb > 1: every camera hit continues)
        # (terminal: the survivors' next segment is the last and a hit there adds nothing — the camera kernel also writes the 16-byte
        # radiance slot such a hit ends with, okrt_fast_trace.hip: FusedArgs::terminalNext)
        term1 = terminal and mb == 1
        cam_alg = (56 + (16 if term1 else 0)) * cam_surv + slot * max(pixels - cam_surv, 0) + 64 * cam_hits + 16 * cam_taps + 16 * cam_env
        cam_l2 = (64 * cam_nodes + 48 * cam_tris) / 64 + cam_alg
        if build.get("flags", 0) & 8:  # OKRT_BUILD_FLAG_CAMERA_PER_RAY: the probe chose the per-ray kernel (three launches, queue 0 and hit records in between)
            cam_alg += (32 + 32 + 20 + 20 + 32) * pixels  # generate writes 32 B per ray, traceKernel reads them + writes 20, shadeKernel<FIRST> reads 32 + 20
            add("traceKernelCamera + generateKernel + shadeKernel<FIRST> (camera segment traced per ray: chosen by the probe for this scene)",
                interval_ms[0] + interval_ms[1], cam_alg, 64 * cam_nodes + 48 * cam_tris + cam_alg, structure_bytes, (32 + 32 + 20) * pixels)
        else:
            add("tracePacketKernel<FUSED> (camera segment: generate + packet traversal + shade + append)", interval_ms[0] + interval_ms[1], cam_alg, cam_l2,
                structure_bytes, 0)
        if mb >= 1 and len(interval_ms) >= 4:
            tr = sum(interval_ms[2::2])
            sh = sum(interval_ms[3::2])
            # traceKernel: 32 B ray read + 20 B hit record write per ray; per-lane gathers of 64 B per node visit, 48 B per triangle test
            # (terminal, NUM_BOUNCES = 1: MODE 1 writes 4 bytes per MISS and nothing per hit; missKernel reads a 4-byte word per ray,
            # 48 B of queue entry + a 16-byte tap + a 16-byte slot per miss. With more bounces only the last segment is reduced and the
            # per-segment ray counts are not known here: the full-record sizes are kept, an over-estimate, and said so.)
            if term1:
                add("traceKernel<MODE 1> (bounce-ray traversal, closest hit; hit-or-miss record)", tr, 32 * bounce_rays + 4 * later_env,
                    64 * b_nodes + 48 * b_tris + 32 * bounce_rays, structure_bytes * mb, 32 * bounce_rays)
                ms_alg = 4 * bounce_rays + (48 + 16 + 16) * later_env
                ms_stream = 4 * bounce_rays
                if frames_per_launch > 1:
                    ms_alg += (16 * frames_per_launch + 36) * pixels / frames_per_launch
                    ms_stream += (16 * frames_per_launch + 16) * pixels / frames_per_launch
                add("missKernel + accumulateKernel (last segment's misses: environment lookup; frame-slot accumulation)", sh, ms_alg, ms_alg, 0, ms_stream)
                tr = sh = 0
            add("traceKernel (bounce-ray traversal)", tr, 52 * bounce_rays, 64 * b_nodes + 48 * b_tris + 52 * bounce_rays, structure_bytes * mb, 32 * bounce_rays)
            later_surv = max(bounce_rays - cam_surv, 0)  # rays of segments 2..: appended by the later shading kernels
            sh_alg = (56 + 20) * bounce_rays + 64 * later_hits + 16 * later_taps + 16 * later_env + slot * max(bounce_rays - later_surv, 0) + 56 * later_surv
            sh_stream = (56 + 20) * bounce_rays
            if frames_per_launch > 1:
                sh_alg += (16 * frames_per_launch + 36) * pixels / frames_per_launch  # accumulateKernel: all slots once + accum RMW + resolve
                sh_stream += (16 * frames_per_launch + 16) * pixels / frames_per_launch
            add("shadeKernel<NEXT> + accumulateKernel (later segments: fetch, lights, BSDF, frame-slot accumulation)", sh, sh_alg, sh_alg, 0, sh_stream)
    elif interval_ms:
        add("renderReferenceOrder (reference-order megakernel)", interval_ms[0], ref_trav_bytes + ref_shade_bytes, ref_trav_bytes + ref_shade_bytes)
    if not kernels:
        return {"bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
    head = max(kernels, key=lambda k: k["launch_ms"])
    pmc, why = load_pmc(args.workload, frames_per_launch) if world == 1 and args.max_bounces == 1 else (None, "PMC files are measured at N = 1, NUM_BOUNCES = 1")
    traffic = lane_util = l2_hit = None
    if pmc:
        fname, j = pmc
        for k in kernels:
            short = k["kernel"].split(" ")[0].split("<")[0]
            m = j.get("kernels", {}).get(short)
            if m:
                k["traffic"] = int(m["hbm_bytes_per_launch"])
                k["traffic_frac"] = round(m["hbm_bytes_per_launch"] / (k["launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                k["l2_hit_pct"] = m.get("l2_hit_pct")
                k["valu_lane_util"] = m.get("valu_lane_util")
                # issue-rate roofline (tools/pmc_traffic.py pass sq2; ceilings measured by tools/calib/issue_rate.hip -> profiles/r05_issue_rate.txt:
                # a SIMD starts at most ONE vector instruction per quad-cycle — min / max / select / compare / convert all measure 0.96-0.98, only
                # v_fma_f32 dual-rates to 1.55-1.77 — and one scalar instruction beside it): instructions of the kind / (SIMDs x kernel quad-cycles)
                # valu_slot_frac (pass sq3) prices the same stream with its measured MIX: slots needed at the per-kind rates / slots the kernel had
                for key in ("valu_issue_frac", "valu_slot_frac", "valu_slot_frac_lower", "valu_dual_rate_share", "valu_busy_frac", "salu_issue_frac", "scalar_busy_frac", "vmem_issue_frac", "lds_issue_frac"):
                    if m.get(key) is not None:
                        k[key] = m[key]
                if m.get("wait_any_share") is not None:
                    k["wait_mem_share"] = m["wait_any_share"]      # wave quad-cycles parked in s_waitcnt / barriers
                    k["wait_issue_share"] = m.get("wait_inst_share")  # ... ready but not issued (the port was taken)
                if m.get("l2_read_requests_per_launch"):
                    req = m["l2_read_requests_per_launch"]
                    k["l2_read_requests_per_launch"] = int(req)
                    # (requests and duration from the SAME profiled launches: the ceiling is itself a measured rate, and a live launch that is
                    # 2 % shorter than the profiled one would otherwise print a fraction above 1)
                    per_s = req / ((m.get("launch_ms") or k["launch_ms"]) * 1e-3)
                    k["l2_frac"] = round(per_s * 128 / 1e9 / L2_PEAK_GBS, 4)
                    # against what a pure gather with the same L2 hit rate reaches — for the kernels whose requests ARE gathers (the shading
                    # kernels also stream their queues, whole 128-byte lines at a higher rate than any gather: the ceiling does not apply)
                    if m.get("l2_hit_pct") is not None and short in ("traceKernel", "tracePacketKernel", "traceKernelCamera"):
                        k["gather_ceiling_frac"] = round(per_s / gather_ceiling(m["l2_hit_pct"] / 100.0), 4)
        traffic, lane_util, l2_hit = head.get("traffic"), head.get("valu_lane_util"), head.get("l2_hit_pct")
        for k in kernels:  # what binds each kernel, from its own measured fractions (the largest one names it)
            cands = {"valu-issue": min(k.get("valu_slot_frac") or k.get("valu_issue_frac") or 0.0, 1.0), "gather (L1 -> L2 request rate of the memory hierarchy)": k.get("gather_ceiling_frac") or 0.0,
                     "fabric (L2 <-> Infinity Cache / HBM bytes)": k.get("traffic_frac") or 0.0}
            best = max(cands, key=lambda n: cands[n])
            k["bound"] = best if cands[best] >= 0.5 else "latency (no resource above half of its ceiling: short dependent launches)"
    measured = traffic is not None
    achieved = (traffic if measured else head["algorithmic_bytes_per_launch"]) / (head["launch_ms"] * 1e-3) / 1e9
    return {
        "bound": "hbm", "kernel": head["kernel"], "launch_ms": head["launch_ms"], "frames_per_launch": frames_per_launch,
        "launches_timed": launches,
        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
        "frac_basis": ("measured fabric-side bytes (traffic) / live launch time / peak; counters calibrated per access pattern, " + (pmc[1].get("calibration") or "uncalibrated")) if measured
                      else "algorithmic bytes (lower bound of HBM traffic) / live launch time / peak",
        "traffic": traffic, "traffic_stale": not measured,
        "traffic_kind": "bytes between L2 and the fabric (Infinity-Cache hits included: an upper bound of HBM bytes for a working set that fits its 256 MiB): "
                        "FETCH_SIZE + half of the coalesced 16-B-per-lane stream reads (the one pattern it counts at half) + WRITE_SIZE" if measured else None,
        "traffic_source": pmc[0] if pmc else why,
        "algorithmic_bytes_per_launch": head["algorithmic_bytes_per_launch"], "algorithmic_frac": head["algorithmic_frac"],
        "l2_frac": head["l2_frac"], "gather_ceiling_frac": head["gather_ceiling_frac"], "l2_hit_pct": l2_hit, "valu_lane_util": lane_util,
        # the dominant kernel's binding resource (kernels[].bound): `bound` above stays the HBM roofline the contract asks for
        "binding_resource": head.get("bound"), "valu_issue_frac": head.get("valu_issue_frac"),
        # the vector port's occupancy priced with the kernel's measured instruction mix, [lower, upper] (tools/pmc_traffic.py pass sq3): full when it brackets 1
        "valu_slot_frac_range": [head.get("valu_slot_frac_lower"), head.get("valu_slot_frac")] if head.get("valu_slot_frac") is not None else None,
        "salu_issue_frac": head.get("salu_issue_frac"),
        "issue_ceiling": "one VALU instruction per SIMD quad-cycle (measured 0.96-0.98 for min / max / select / compare / convert; v_fma_f32 alone 1.55-1.77) "
                         "and one SALU instruction beside it: tools/calib/issue_rate.hip -> profiles/r05_issue_rate.txt",
        "kernels": kernels,
        "node4q_visits_per_bounce_ray": round(b_nodes / max(bounce_rays, 1), 2), "tri48_tests_per_bounce_ray": round(b_tris / max(bounce_rays, 1), 2),
        # (wave-level counts under the packet kernel, per packet of 64 rays; under the per-ray camera kernel the same expression is the average per RAY)
        "node4q_steps_per_camera_tile": round(cam_nodes / 64 / max(pixels / 64, 1), 2), "tri48_tests_per_camera_tile": round(cam_tris / 64 / max(pixels / 64, 1), 2),
        "camera_segment": "per ray (probe)" if build.get("flags", 0) & 8 else "packets of 64 rays: 4x2 pixels x 8 frames (4x4 x 4, then 8x8 x 1, for what a launch set has beyond a multiple of eight / four frames)",
        "structure_bytes": structure_bytes,
        "traversal_ms_per_frame": round(trace_ms_frame, 4), "later_shading_ms_per_frame": round(shade_ms_frame, 4), "other_ms_per_frame": round(other_ms_frame, 4),
        # SURVEY.md §8d to the letter, kept under its own name: the bytes the REFERENCE's octree -> mesh -> BVH order would touch
        # for these rays over the traversal time. Far above 1: it measures the work the derived structure avoids, not a bandwidth.
        "reference_work_ratio": round(ref_trav_bytes / per_gpu / max(trace_ms_frame * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 2),
        "reference_algorithmic_bytes_per_ray": round(ref_trav_bytes / max(ref_c["segments"], 1), 1),
        "note": "frac <= 1 by construction. kernels[].bound names what binds each kernel: on a cache-resident structure the traversal kernels fill the SIMDs' vector "
                "issue port (valu_issue_frac ~ 1 of one instruction per SIMD quad-cycle, at valu_lane_util of the lanes), so their fabric fraction is low; on structures "
                "beyond the caches (871 k / 8 M triangles) they run at the memory hierarchy's gather rate (gather_ceiling_frac ~ 1); the streaming kernels move the most "
                "bytes per second (traffic_frac), a fabric / Infinity-Cache rate on this working set, not proven HBM.",
    }


def device_memory_in_use_gb(devices):
    """Largest bytes in use over the job's devices right now (hipMemGetInfo: total - free; everything on the device counts,
    other processes included), or None when the runtime cannot say."""
    try:
        hip = C.CDLL("libamdhip64.so.7")  # the soname libokrt.so is linked against: the runtime instance already loaded
        worst = 0
        for d in sorted(set(devices)):
            free, total = C.c_size_t(0), C.c_size_t(0)
            if hip.hipSetDevice(C.c_int(d)) != 0 or hip.hipMemGetInfo(C.byref(free), C.byref(total)) != 0:
                return None
            worst = max(worst, total.value - free.value)
        return round(worst / 1e9, 2)
    except OSError:
        return None


class RankState:
    """One rank this process drives: its RayTracer (host layer) on its device, with the rank's tile."""

    def __init__(self, rank, device, rt, scene, res):
        self.rank, self.device, self.rt, self.scene, self.res = rank, device, rt, scene, res
        self.ctx = rt.ctx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="sponza", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-call", action="store_true", help="skip the one-okrt_render-per-frame and 1024-spp timings (profiling runs: every launch set then has frames_per_launch frames)")
    ap.add_argument("--no-repeats", action="store_true", help="skip the four untimed-for-the-headline repeats of the K-step region (profiling runs)")
    ap.add_argument("--traversal", default="fast", choices=["fast", "reference"])
    ap.add_argument("--max-bounces", type=int, default=1, help="NUM_BOUNCES (RaytracerCS.hlsl:6; HEAD = 1, used for every graded number)")
    ap.add_argument("--last-segment", default="shortcut", choices=["shortcut", "full", "any-hit"],
                    help="what the fast pipeline does on the last path segment where a hit there provably adds +-0 (no emitters, no lights): "
                         "'shortcut' (default, OKRT_OPT_LAST_SEGMENT_SHORTCUT = 1: hit-or-miss record, only misses shaded; bit-identical), "
                         "'full' (every segment shaded in full), 'any-hit' (developer knob: OKRT_OPT_LAST_SEGMENT_ANY_HIT, not closest-hit rays)")
    ap.add_argument("--split", default="tiles", choices=["tiles", "samples"],
                    help="N > 1: 'tiles' = image-space stripes + gather (the north-star path); 'samples' = every rank renders "
                         "the whole frame for its own block of frame indices, accumulations are summed with one reduce")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="consecutive frames traced by one set of kernel launches (okrt_render_frames); 0 = auto")
    args = ap.parse_args()
    # stdout carries ONE JSON line and nothing else: libraries that greet on stdout (RCCL prints a version banner from its
    # first communicator) are sent to stderr for the length of the run; the line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    # the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails with
    # "hipIpcGetMemHandle: invalid argument" (must be in the environment before the HIP runtime initialises)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    abi = importlib.import_module("gpu-raytracer_amd._abi")
    scenes = importlib.import_module("gpu-raytracer_amd.scenes")
    mgpu = importlib.import_module("gpu-raytracer_amd.mgpu")
    # who drives which rank: decided from the command line and the environment BEFORE anything touches a GPU
    #   plain `python bench.py --gpus N`       -> this one process drives N contexts (C-ABI exchange, no launcher needed)
    #   under a launcher (WORLD_SIZE / RANK)   -> one process per rank; the launcher's world size wins over --gpus
    plan = mgpu.launch_plan(args.gpus, os.environ, None)
    fns = abi.load_device_library()  # raises if libokrt.so is missing: no fallback
    plan = mgpu.launch_plan(args.gpus, os.environ, mgpu.visible_gpus(fns))
    world, rank = plan["world"], plan["rank"]
    args.gpus = world
    store = mgpu.FileStore(rank, world) if plan["mode"] == "process_per_gpu" else None
    plan = mgpu.settle_shared_devices(plan, store, os.environ)  # ranks that turn out to share a GPU: rehearsal

    builder, kwargs, W, H, wl_desc = WORKLOADS[args.workload]
    desc = getattr(scenes, builder)(**kwargs)

    # scene tables are replicated: every rank's context gets the same uploads through the host layer
    def make_rank(r_dev):
        r, dev = r_dev
        rt, scene, res = scenes.instantiate(desc, W, H, device=dev)
        return RankState(r, dev, rt, scene, res)
    importlib.import_module("gpu-raytracer_amd.host").host_library()  # loaded once, before any worker thread needs it
    if len(plan["local_ranks"]) > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(plan["local_ranks"]), usable_cpus())) as pool:
            ranks = list(pool.map(make_rank, zip(plan["local_ranks"], plan["devices"])))
    else:
        ranks = [make_rank((plan["local_ranks"][0], plan["devices"][0]))]
    rt, ctx = ranks[0].rt, ranks[0].ctx  # the rank whose kernels the roofline reports (rank 0 in one_process / single mode)
    samples_split = world > 1 and args.split == "samples"
    for st in ranks:
        st.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_FAST if args.traversal == "fast" else abi.TRAVERSAL_REFERENCE)
        st.ctx.set_option(abi.OPT_MAX_BOUNCES, args.max_bounces)
        st.ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, 0 if args.last_segment == "full" else 1)
        st.ctx.set_option(abi.OPT_LAST_SEGMENT_ANY_HIT, 1 if args.last_segment == "any-hit" else 0)
        if world > 1 and not samples_split:
            st.ctx.set_tile(STRIPE_ROWS, st.rank, world)
    if world == 1 and os.environ.get("OKRT_BENCH_TILE"):  # developer knob: time one rank's share of an N-way split on 1 GPU
        sr, si, ss = (int(v) for v in os.environ["OKRT_BENCH_TILE"].split(","))
        ctx.set_tile(sr, si, ss)
    # the root of the exchanges: the rank that owns the fewest rows of the striped split (it finishes its own frames first); the
    # samples split has no lighter rank
    group = mgpu.TileGroup(fns, plan, [st.ctx.h for st in ranks], store, frame=(W, H, STRIPE_ROWS), read_tile=lambda k: ranks[k].ctx.read_accum(),
                           root=0 if samples_split else None)

    def exchange(final=False):
        if world <= 1:
            return
        if samples_split:  # disjoint frame indices, same pixels: one sum at the very end
            if final:
                group.reduce()
            return
        group.gather()

    def sync(before_region=False):
        for st in ranks:
            st.ctx.synchronize()
        if store:
            # leaves all ranks within microseconds of each other and adds microseconds itself (the file-polling barrier: up to 2 ms,
            # the length of an 8-GPU 20-frame region); in front of a region the ranks first collect asleep
            store.tight_barrier(gather_first=before_region)

    # ---- roofline numerator: the reference-order counters of one frame of this workload -------
    def counted(st, traversal, max_bounces, frames=1):
        """okrt_counters of `frames` frames in one launch set, per frame. The fast pipeline is counted over EIGHT frames: its camera kernel
        cuts a launch set's packets as 4x2 pixels x 8 frames (frame-stacked packets; 4x4 x 4 and 8x8 x 1 for what is left beyond a multiple
        of eight / four frames), which is the shape most of the timed region runs."""
        st.ctx.set_option(abi.OPT_TRAVERSAL, traversal)
        st.ctx.set_option(abi.OPT_MAX_BOUNCES, max_bounces)
        st.rt.resetAccumulation()
        st.ctx.reset_counters()
        st.rt.renderFrames(frames)
        c = st.ctx.counters()
        return {k: (v / frames if frames > 1 else v) for k, v in c.items()}

    def add_counters(dicts):
        return {k: sum(d[k] for d in dicts) for k in dicts[0]}

    for st in ranks:
        st.ctx.set_option(abi.OPT_COUNTERS, 1)
    ref_counters = add_counters([counted(st, abi.TRAVERSAL_REFERENCE, args.max_bounces) for st in ranks])
    fast_counters = add_counters([counted(st, abi.TRAVERSAL_FAST, args.max_bounces, 8) for st in ranks])  # its own node / triangle visit counts, same frame indices
    # the camera segment alone (max_bounces = 0): splits the node / triangle visit counts between the two traversal kernels
    camera_counters = add_counters([counted(st, abi.TRAVERSAL_FAST, 0, 8) for st in ranks])
    for st in ranks:
        st.ctx.set_option(abi.OPT_MAX_BOUNCES, args.max_bounces)
        st.ctx.set_option(abi.OPT_COUNTERS, 0)
        st.ctx.set_option(abi.OPT_TRAVERSAL, abi.TRAVERSAL_FAST if args.traversal == "fast" else abi.TRAVERSAL_REFERENCE)
        st.rt.resetAccumulation()

    # ---- warm-up, then exactly K timed steps ----------------------------------------------------
    # A step is one frame. Frames are handed to the device layer F at a time (okrt_render_frames): same framebuffers
    # bit for bit, but one set of kernel launches traces F frames' rays, which amortises the per-launch latency floor
    # (the longest ray's dependent steps) — decisive when a rank only owns 1/N of the pixels.
    # default: 64 frames per launch set (= the gather period with N ranks)
    fpl = args.frames_per_launch or int(os.environ.get("OKRT_BENCH_FPL", "0")) or 64
    gather_every = int(os.environ.get("OKRT_BENCH_GATHER_EVERY", GATHER_EVERY))
    fpl = max(1, min(fpl, 64, gather_every))
    enqueue_s = [0.0, 0]  # host time spent inside renderFrames calls (they must only ENQUEUE: that is what lets one thread feed N GPUs)

    # one process, several GPUs: every rank's launch set is enqueued from its own thread (the call is one ctypes call into the C++
    # host layer, which releases the interpreter lock), so rank N-1's GPU starts with rank 0's instead of N-1 enqueue times later
    enqueue_pool = None
    if len(ranks) > 1:
        from concurrent.futures import ThreadPoolExecutor
        enqueue_pool = ThreadPoolExecutor(max_workers=len(ranks))

    def enqueue(st, m):
        t_enq = time.perf_counter()
        st.rt.renderFrames(m)
        return time.perf_counter() - t_enq

    def run_frames(n, gather):
        done = 0
        while done < n:
            m = min(fpl, n - done, gather_every - done % gather_every)
            # asynchronous launches: the GPUs of a one-process job run concurrently
            spent = list(enqueue_pool.map(lambda st: enqueue(st, m), ranks)) if enqueue_pool else [enqueue(ranks[0], m)]
            enqueue_s[0] += sum(spent)
            enqueue_s[1] += len(spent)
            done += m
            if gather and done % gather_every == 0 and done != n:
                exchange()

    run_frames(args.warmup, False)
    # also run one full-size launch set untimed unless the warm-up already did: its queues (GBs) are allocated on first use
    extra_warm = fpl if (args.warmup == 0 or args.warmup % fpl) else 0
    run_frames(extra_warm, False)
    # ... and the launch-set shapes of the timed region itself, twice: a K-step region that is not a multiple of the set size ends
    # in a smaller set, small tiles issue their sets as sub-sets that take turns on four queue slots (csrc/okrt_launch_plan.h), and
    # every slot allocates its queues on first use — 2.6 ms per rank inside a 1.7 ms region when this was missing (r04 N = 8 rehearsal)
    k_here = args.steps // world if (world > 1 and args.split == "samples") else args.steps
    shape_warm = k_here if k_here <= fpl else fpl + k_here % fpl
    # (the per-kernel event ring is switched on BEFORE these rehearsals: its events are created on a slot's first use, and the first
    # timed region used to pay for that — it read 1-8 % above its four repeats, the more the smaller the rank's tile)
    ctx.set_option(abi.OPT_TIMING, min(5 * args.steps, 4096))
    for _ in range(2):
        run_frames(shape_warm, False)
    if world > 1 and not samples_split:
        exchange()  # untimed: communicator warm-up (RCCL connects its channels on first use), staging buffers allocated
    steps_here = args.steps
    if samples_split:  # rank r renders frame indices r*K/N + 1 ... (r+1)*K/N of a fresh accumulation
        assert args.steps % world == 0, "--split samples needs steps divisible by the number of ranks"
        steps_here = args.steps // world
        for st in ranks:
            st.rt.resetAccumulation()
            st.rt.render_data.numAccumulationFrames = st.rank * steps_here
    sync()
    for st in ranks:
        st.ctx.reset_counters()
    ctx.set_option(abi.OPT_TIMING, min(5 * args.steps, 4096))  # HIP events around every kernel of every launch set, on the stream they run on (ring: all five regions)
    enqueue_s[0], enqueue_s[1] = 0.0, 0
    sync(before_region=True)
    t0 = time.perf_counter()
    for st in ranks:
        st.ctx.timer_begin()
    run_frames(steps_here, not samples_split)
    for st in ranks:
        st.ctx.timer_end()  # every rank's own span: its frames, without the final exchange
    exchange(final=True)
    sync()
    elapsed = time.perf_counter() - t0
    rank_region = {st.rank: st.ctx.timer_elapsed_ms() for st in ranks}
    gather_in_region = group.exchange_ms() if world > 1 else None  # root only: device time of the region's last exchange on the root's stream
    mem_gb = device_memory_in_use_gb(plan["devices"])  # queues of the timed launch sets are allocated now
    enqueue_ms_per_set = enqueue_s[0] * 1e3 / max(enqueue_s[1], 1)
    trace_ms, shade_ms, other_ms, timed_frames = ctx.stage_ms()
    interval_ms, interval_launches, interval_frames = ctx.interval_ms()  # per kernel, launch sets of the size of the last one
    build = ctx.build_info()
    rays_local = sum(st.ctx.counters()["segments"] for st in ranks)
    # The graded region above is exactly K steps, once. The same region four more times says how far one sample can be trusted
    # (config.ms_per_step_repeats; the headline stays the first) and gives the per-kernel averages below five regions to stand on.
    repeat_ms = [elapsed * 1e3 / args.steps]
    repeats_done = 0
    if not samples_split and not args.no_repeats:
        for _ in range(4):
            sync(before_region=True)
            t1 = time.perf_counter()
            run_frames(steps_here, True)
            exchange(final=True)
            sync()
            repeat_ms.append((time.perf_counter() - t1) * 1e3 / args.steps)
            repeats_done += 1
        interval_ms, interval_launches, interval_frames = ctx.interval_ms()  # all five regions' launch sets (same shapes)
    ctx.set_option(abi.OPT_TIMING, 0)
    # Does the last-segment shortcut apply to this workload? (okrt.h: OKRT_OPT_LAST_SEGMENT_SHORTCUT — materials without emission and
    # with |albedo| <= 1, no light in the frame's RenderData.) If it does, the same K-step region is also timed with every segment
    # shaded in full and with any-hit on the last segment, under their own names; the graded line stays the closest-hit default.
    rd_now = rt.render_data
    shortcut_applies = (args.traversal == "fast" and args.max_bounces >= 1 and bool(build["flags"] & abi.BUILD_FLAG_DARK_MATERIALS)
                        and rd_now.numDirLights == 0 and rd_now.numPointLights == 0 and rd_now.numSpotLights == 0)
    terminal = shortcut_applies and args.last_segment != "full"
    alt_ms = {}
    if world == 1 and shortcut_applies and not args.no_single_call:
        for name, (opt_short, opt_any) in (("full_shading", (0, 0)), ("any_hit", (1, 1))):
            ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, opt_short)
            ctx.set_option(abi.OPT_LAST_SEGMENT_ANY_HIT, opt_any)
            run_frames(shape_warm, False)
            best = None
            for _ in range(3):
                sync(before_region=True)
                t1 = time.perf_counter()
                run_frames(steps_here, True)
                sync()
                ms = (time.perf_counter() - t1) * 1e3 / args.steps
                best = ms if best is None else min(best, ms)
            alt_ms[name] = best
        ctx.set_option(abi.OPT_LAST_SEGMENT_SHORTCUT, 0 if args.last_segment == "full" else 1)
        ctx.set_option(abi.OPT_LAST_SEGMENT_ANY_HIT, 1 if args.last_segment == "any-hit" else 0)
    if args.traversal == "reference":  # the reference-order kernel only counts with OKRT_OPT_COUNTERS (slow): segments of the counted frame x frames
        rays_local = ref_counters["segments"] * steps_here
    gpu_event_ms = ctx.timer_elapsed_ms()

    # the combined frame on rank 0: every pixel must have received every frame
    frames_done = args.steps if samples_split else args.warmup + extra_warm + 2 * shape_warm + args.steps * (1 + repeats_done)
    if world > 1:
        full = group.read_full()
        if full is not None:
            assert bool((full[..., 3] == frames_done).all()), "combined frame: accumulation alpha != number of frames"
    # one exchange on its own (all ranks idle before it starts): what a gather costs when nothing hides it
    exchange_ms = None
    if world > 1:
        sync()
        t1 = time.perf_counter()
        exchange(final=True)
        sync()
        exchange_ms = (time.perf_counter() - t1) * 1e3

    # for reference: the same frames issued one okrt_render call per frame (the reference's main-loop shape), and the
    # configuration BASELINE.json names as such: 1024 spp of progressive accumulation, end to end
    single_ms = single_sync_ms = spp1024_ms = readback_ms = single_frame_ms = None
    if world == 1 and not args.no_single_call:
        def per_call(defer):
            ctx.set_option(abi.OPT_DEFER, defer)
            ctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(64):
                rt.render()
            ctx.synchronize()
            return (time.perf_counter() - t1) * 1e3 / 64
        # SURVEY 8d "ms/frame for ONE 1-spp frame": one frame launched and waited for, nothing deferred, nothing else in flight
        ctx.set_option(abi.OPT_DEFER, 0)
        ctx.set_option(abi.OPT_PIPELINE, 0)
        lat = []
        for _ in range(17):
            ctx.synchronize()
            t1 = time.perf_counter()
            rt.render()
            ctx.synchronize()
            lat.append((time.perf_counter() - t1) * 1e3)
        single_frame_ms = sorted(lat[1:])[8]  # median of 16 (the first pays the single-frame queue shape)
        ctx.set_option(abi.OPT_PIPELINE, 4)
        single_sync_ms = per_call(0)   # every okrt_render launches its own frame
        single_ms = per_call(16)       # the default: the library defers and batches the accumulation loop's frames
        rt.resetAccumulation()
        ctx.synchronize()
        t1 = time.perf_counter()
        done = 0
        while done < 1024:
            rt.renderFrames(min(fpl, 1024 - done))
            done += min(fpl, 1024 - done)
        ctx.synchronize()
        spp1024_ms = (time.perf_counter() - t1) * 1e3 / 1024
        readback_ms = 1e9
        for _ in range(3):  # (the first call pays the first touch of its destination pages; a display loop pays the steady cost)
            t1 = time.perf_counter()
            img = ctx.read_result()  # the resolve is part of every launch set; this is the copy of the RGBA8 result to the host
            readback_ms = min(readback_ms, (time.perf_counter() - t1) * 1e3)
        assert img.shape == (H, W, 4)

    # ---- job-wide sums and maxima (one_process: all ranks are local) -----------------------------
    def job_sum(v):
        return mgpu.allreduce_json(store, v, lambda xs: sum(xs))

    def job_max(v):
        return mgpu.allreduce_json(store, v, lambda xs: max(xs))

    def job_sum_dict(d):
        return mgpu.allreduce_json(store, d, lambda ds: {k: sum(x[k] for x in ds) for k in ds[0]})
    if store:  # one process per rank: every rank's span and the root's exchange time travel through the store
        merged = {}
        for part in [json.loads(b.decode()) for b in store.allgather(json.dumps({str(k): v for k, v in rank_region.items()}).encode())]:
            merged.update({int(k): v for k, v in part.items()})
        rank_region = merged
        gather_in_region = job_max(gather_in_region if gather_in_region is not None else -1.0)
        gather_in_region = None if gather_in_region is None or gather_in_region < 0 else gather_in_region
    rays_total = job_sum(rays_local)
    elapsed_max = job_max(elapsed)
    repeat_ms = [job_max(v) for v in repeat_ms]
    if store:
        ref_counters, fast_counters, camera_counters = job_sum_dict(ref_counters), job_sum_dict(fast_counters), job_sum_dict(camera_counters)
        trace_ms, shade_ms, other_ms = job_max(trace_ms), job_max(shade_ms), job_max(other_ms)
        exchange_ms = job_max(exchange_ms)

    if rank == 0:
        ms_per_step = elapsed_max * 1e3 / args.steps
        mrays = rays_total / elapsed_max / 1e6
        ref_trav_bytes, ref_shade_bytes = algorithmic_bytes(ref_counters)
        frames = max(timed_frames, 1)
        trace_ms_frame = trace_ms / frames
        fpl_used = max(1, min(fpl, steps_here))  # the launch-set size the timed region really used
        tile = {"rows": H, "bytes": H * W * 16}
        if world > 1 and not samples_split:
            tile = {"rows": ranks[0].ctx.tile_rows(), "bytes": ranks[0].ctx.tile_rows() * W * 16}
        if world == 1:
            gather_desc = "none"
        elif samples_split:
            gather_desc = f"one sum of the whole-frame float4 accumulations onto rank 0 after the last frame ({group.transport})"
        else:
            how = {"rccl": "RCCL point-to-point (one ncclSend per peer, one ncclRecv per peer on rank 0, one group) over xGMI",
                   "copy": "REHEARSAL copy transport (ranks share a GPU: device copies instead of RCCL)"}.get(group.transport, group.transport)
            gather_desc = f"{how}: packed float4 tiles to rank 0 every {gather_every} frames + final, de-interleaved there by one kernel"
        out = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl_desc, "traversal": args.traversal, "max_bounces": args.max_bounces, "rays_per_frame": rays_total // args.steps, "frames_per_launch": fpl_used,
                       "gpu_event_ms_per_step": round(gpu_event_ms / max(steps_here, 1), 4),
                       # the graded K-step region (first value = ms_per_step) and the same region repeated four more times
                       "ms_per_step_repeats": [round(v, 4) for v in repeat_ms],
                       "ms_per_step_repeats_spread": round((max(repeat_ms) - min(repeat_ms)) / min(repeat_ms), 4),
                       "last_segment": ("any-hit on the last segment (OKRT_OPT_LAST_SEGMENT_ANY_HIT): NOT closest-hit rays, developer knob, not a graded number" if terminal and args.last_segment == "any-hit" else
                                        "closest-hit traversal, hit-or-miss record, only misses shaded: no emitters and no lights in this scene, so a last-segment hit adds +-0 "
                                        "(OKRT_OPT_LAST_SEGMENT_SHORTCUT, default, framebuffers bit-identical to full shading)" if terminal else
                                        "shaded in full" + ("" if not shortcut_applies else " (--last-segment full)")),
                       "ms_per_step_last_segment_full_shading": round(alt_ms["full_shading"], 4) if "full_shading" in alt_ms else None,
                       "ms_per_step_last_segment_any_hit": round(alt_ms["any_hit"], 4) if "any_hit" in alt_ms else None,
                       "ms_single_frame": round(single_frame_ms, 4) if single_frame_ms else None,
                       "ms_per_step_one_call_per_frame": round(single_ms, 4) if single_ms else None,
                       "ms_per_step_one_call_per_frame_defer_off": round(single_sync_ms, 4) if single_sync_ms else None,
                       "ms_per_step_1024spp": round(spp1024_ms, 4) if spp1024_ms else None,
                       "read_result_rgba8_ms": round(readback_ms, 3) if readback_ms else None,
                       "launch": {"single": "one process, one context", "one_process": f"one process drives {world} contexts through the C ABI (no launcher, no torch)",
                                  "process_per_gpu": f"one process per rank ({world}), RCCL id handed over through a file store (no torch)"}[plan["mode"]],
                       "rehearsal": bool(plan["rehearsal"]), "devices": plan["devices"] if plan["mode"] != "process_per_gpu" else None,
                       "tile_split": ("samples split: every rank renders the whole frame for its own block of frame indices" if samples_split else
                                      f"{STRIPE_ROWS}-row stripes interleaved over {world} ranks") if world > 1 else "full frame",
                       "gather": gather_desc, "gather_transport": group.transport if world > 1 else None,
                       "gather_root": group.root if world > 1 else None,
                       # device time of every rank's own frames of the timed region (HIP events on its stream), in rank order; the N-GPU
                       # job's time is the slowest of them plus what the exchange adds behind it
                       "rank_region_ms": [round(rank_region[r], 4) for r in sorted(rank_region)] if world > 1 else None,
                       "slowest_rank": max(rank_region, key=lambda r: rank_region[r]) if world > 1 else None,
                       # the region's last exchange on the root's stream, from "the root's own frames are done" to the assembled frame
                       # (okrt_comm_exchange_ms): the wait for slower peers + transfers + de-interleave
                       "gather_ms_in_region": round(gather_in_region, 4) if gather_in_region is not None else None,
                       "gather_ms": round(exchange_ms, 3) if exchange_ms is not None else None,
                       "gather_bytes_per_peer": tile["bytes"] if world > 1 else None,
                       "host_enqueue_ms_per_launch_set": round(enqueue_ms_per_set, 3),
                       "device_memory_in_use_gb": mem_gb},
        }
        if plan["rehearsal"]:
            out["config"]["note"] = (f"REHEARSAL: {world} ranks on fewer GPUs (devices {plan['devices']}); the ranks share GPU time, so this line says "
                                     "nothing about scaling — it exercises the N-rank code path")
        elif world > 1 and "FALLBACK" in group.transport:
            out["config"]["note"] = ("the RCCL communicator could not be created (gather_transport has the error): the tiles were exchanged through "
                                     "host memory, so gather_ms is not an xGMI figure; value covers the ranks' rendering only, as always")
        out["roofline"] = roofline_report(args, world, samples_split, interval_ms, interval_launches, interval_frames, build,
                                          ref_counters, fast_counters, camera_counters, ref_trav_bytes, ref_shade_bytes, trace_ms_frame,
                                          shade_ms / frames, other_ms / frames, terminal)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, W, H, None)
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    if enqueue_pool:
        enqueue_pool.shutdown()
    group.close()
    for st in ranks:
        st.rt.close()
    if store:
        store.close()


if __name__ == "__main__":
    main()
