#!/usr/bin/env python3
"""Measured per-kernel numbers for bench.py's roofline object, from rocprofv3 passes of the SAME bench command.

  python3 tools/pmc_traffic.py --workload sponza --frames 64 --out gpurun_out/pmc    (on the GPU box, via gpurun)

Five separate passes (counters never share a run with --stats; FETCH_SIZE and WRITE_SIZE do not fit one pass:
MI355X_MICROARCH.md "rocprofv3 PMC slots"):
  1. --kernel-trace --stats                     per-kernel launch durations (must agree with bench.py's HIP-event times)
  2. --pmc FETCH_SIZE                           L2 -> fabric read requests x 64 B (KiB). Calibrated against known bytes per access pattern
                                                (tools/pmc_calibrate.py -> profiles/r03_pmc_calibration.txt): exact for 64-byte gathers, 48-byte
                                                records and texel taps (the over-fetch it shows is real 64-byte granularity), HALF for coalesced
                                                16-B-per-lane streams (128-byte requests tallied at 64) -> half of the kernel's stream reads is added back
  3. --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum  fabric write bytes (KiB, exact), L2 hit rate
  4. --pmc SQ_* (8 slots)                       VALU lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU), wait shares
  5. --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum   read requests the vector L1s send to L2 (128-byte lines: tools/calib/l1_gather_rate.hip
                                                counts exactly one per random 64-byte record and 1.25-1.3 per 48-byte record) and the per-lane-instruction
                                                accesses the L1s serve: how much of a traversal's gathering L1 absorbs, and how far the kernel is from the
                                                request rate a pure gather reaches (profiles/r03_l1_gather_rate.txt)
Every bench run uses --steps F --warmup F --frames-per-launch F, so every launch of the non-counting kernel variants
traces exactly F frames and "per launch" is well defined.

Writes <out>/pmc_<workload>_f<F>.json (+ .txt), stamped with the kernel-source fingerprint bench.py checks before it
attaches `traffic` to a bench line; tools/stamp_profile.py copies it under profiles/ with the git head.
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

PASSES = [
    ("stats", ["--kernel-trace", "--stats"]),
    ("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"]),
    ("write", ["--pmc", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum", "--kernel-trace"]),
    ("sq", ["--pmc", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
            "SQ_WAIT_INST_ANY", "--kernel-trace"]),
    ("l1", ["--pmc", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "--kernel-trace"]),
    # issue-rate roofline (round 5): what the scalar unit, the vector-memory and the LDS instructions take besides the VALU of pass `sq`,
    # and the clock: GRBM_GUI_ACTIVE = busy cycles summed over the 8 XCDs (GRBM has its own 2 slots: it rides this pass)
    ("sq2", ["--pmc", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_SCA", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_ANY",
             "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "--kernel-trace"]),
    # the instruction MIX of the vector stream: fp32 add / mul / fma start at 1.75-1.77 per SIMD quad-cycle, everything else at 0.96-0.98
    # (selects, compares, min / max, converts, most integer forms), transcendentals take two slots (profiles/r05_issue_rate.txt) — with the
    # mix the port's occupancy can be priced per kernel (valu_slot_frac <= ~1) instead of read off an instruction count whose ceiling moves
    ("sq3", ["--pmc", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT",
             "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "--kernel-trace"]),
]
DUAL_RATE = 1.76  # fp32 add / mul / fma per SIMD quad-cycle, measured (tools/calib/issue_rate.hip)
FAMILIES = ("tracePacketKernel", "traceKernel", "shadeKernel", "missKernel", "accumulateKernel", "generateKernel", "renderReferenceOrder")
SIMDS = 256 * 4  # MI355X: 256 CUs x 4 SIMDs


def family(name):
    """(family, counting variant?) of a kernel name; None for kernels that are not part of the frame."""
    m = re.search(r"(tracePacketKernel|traceKernel|shadeKernel|missKernel|accumulateKernel|generateKernel|renderReferenceOrder)(<([^>]*)>)?", name)
    if not m:
        return None, False
    args = [a.strip() for a in (m.group(3) or "").split(",")] if m.group(3) else []
    fam = m.group(1)
    counting = (fam in ("tracePacketKernel", "traceKernel") and args[:1] == ["true"]) or (fam == "shadeKernel" and args[1:2] == ["true"])
    # traceKernel<COUNT, MODE>: MODE 0 = full hit record, 1 / 2 = the last segment's hit-or-miss form. A launch set that holds both is one
    # whose CAMERA segment is traced per ray (MODE 0) in front of the last segment (MODE 1): kept apart, renamed in main()
    if fam == "traceKernel" and args[1:2] == ["0"]:
        fam = "traceKernel0"
    return fam, counting


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="sponza")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default="gpurun_out/pmc")
    ap.add_argument("--passes", default="stats,fetch,write,sq,l1,sq2,sq3")
    args = ap.parse_args()
    import bench
    args.out = os.path.abspath(args.out)  # rocprofv3 runs from /tmp
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    tag = f"{args.workload}_f{args.frames}"
    cmd = ["python3", os.path.join(REPO, "bench.py"), "--workload", args.workload, "--steps", str(args.frames), "--warmup", str(args.frames),
           "--frames-per-launch", str(args.frames), "--no-cpu-baseline", "--no-single-call", "--no-repeats"]
    bench_lines = {}
    for name, flags in PASSES:
        if name not in args.passes.split(","):
            continue
        d = os.path.join(args.out, f"{tag}_{name}")
        log = os.path.join(args.out, f"{tag}_{name}.log")
        print(f"[pmc_traffic] pass {name}: rocprofv3 {' '.join(flags)}", flush=True)
        with open(log, "w") as lf:
            subprocess.run(["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "--"] + cmd, stdout=lf, stderr=subprocess.STDOUT, env=env, cwd="/tmp", check=True)
        js = [l for l in open(log) if l.startswith("{")]
        if js:
            bench_lines[name] = json.loads(js[-1])

    # durations from the stats pass's kernel trace
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(args.out, f"{tag}_stats", "*", "*_kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            fam, counting = family(r["Kernel_Name"])
            if fam and not counting:
                dur[fam].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("fetch", "write", "sq", "l1", "sq2", "sq3"):
        for f in glob.glob(os.path.join(args.out, f"{tag}_{name}", "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                fam, counting = family(r["Kernel_Name"])
                if fam and not counting:
                    # (SQ_INSTS_VALU rides both SQ passes: pass sq2's copy sits beside GRBM_GUI_ACTIVE, i.e. the same launches' clock)
                    cname = r["Counter_Name"] + ("@" + name if name in ("sq2", "sq3") and r["Counter_Name"] in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE") and
                                                  not (name == "sq2" and r["Counter_Name"] == "GRBM_GUI_ACTIVE") else "")
                    ctr[fam][cname].append(float(r["Counter_Value"]))
    for table in (dur, ctr):  # (see family(): MODE 0 beside MODE 1 = the per-ray camera segment; MODE 0 alone = every bounce segment)
        if "traceKernel0" in table:
            table["traceKernelCamera" if "traceKernel" in table else "traceKernel"] = table.pop("traceKernel0")
    kernels = {}
    for fam in FAMILIES + ("traceKernelCamera",):
        if fam not in dur and fam not in ctr:
            continue
        avg = {c: sum(v) / len(v) for c, v in ctr[fam].items()}
        # a family may be launched more than once per launch set (traceKernel / shadeKernel with max_bounces > 1): per launch SET
        per_set = 1
        fetch, write = avg.get("FETCH_SIZE", 0.0) * 1024.0, avg.get("WRITE_SIZE", 0.0) * 1024.0
        hit, miss = avg.get("TCC_HIT_sum", 0.0), avg.get("TCC_MISS_sum", 0.0)
        # the kernel's coalesced 16-B-per-lane stream reads (bench.py knows them from the ray / pixel counts): counted at half
        stream = 0.0
        for bk in ((bench_lines.get("stats") or {}).get("roofline") or {}).get("kernels", []):
            if bk["kernel"].split(" ")[0].split("<")[0] == fam:
                stream = float(bk.get("stream_read_bytes_per_launch", 0))
        add_back = min(stream, 2.0 * fetch) / 2.0
        k = {"launches_seen": len(dur.get(fam, [])), "launch_ms": round(sum(dur[fam]) / len(dur[fam]), 4) if dur.get(fam) else None,
             "fetch_bytes_raw": fetch * per_set, "write_bytes": write * per_set, "stream_read_bytes": stream, "stream_add_back_bytes": add_back,
             "hbm_bytes_per_launch": (fetch + add_back + write) * per_set,
             "l2_hit_pct": round(100.0 * hit / (hit + miss), 2) if hit + miss else None}
        if avg.get("SQ_ACTIVE_INST_VALU"):
            k["valu_lane_util"] = round(avg["SQ_THREAD_CYCLES_VALU"] / (64.0 * avg["SQ_ACTIVE_INST_VALU"]), 4)
            wc = avg.get("SQ_WAVE_CYCLES", 0.0)
            if wc:
                k["wait_any_share"] = round(avg.get("SQ_WAIT_ANY", 0.0) / wc, 4)
                k["wait_inst_share"] = round(avg.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
                k["valu_inst_per_wave_cycle"] = round(avg.get("SQ_INSTS_VALU", 0.0) / wc, 4)
        # Issue-rate roofline. A SIMD gets one issue slot per quad-cycle (4 shader clocks) and can start at most one instruction of
        # each kind in it, so over a kernel of Q quad-cycles (GRBM_GUI_ACTIVE / 8 XCDs / 4) the chip has SIMDS * Q slots per kind:
        #   *_issue_frac = instructions of that kind / slots        (how full the kind's issue port is)
        #   *_busy_frac  = SQ_ACTIVE_INST_* / slots                 (quad-cycles some wave spent executing that kind; VALUBusy / SALUBusy
        #                                                            of the gfx9 derived-counter files; an instruction may hold its unit for
        #                                                            more than one slot: transcendentals 2, wave64 DP more)
        # Both passes' duration may differ slightly from an un-profiled launch (the guide's DVFS note): fractions, not times, are kept.
        gui = avg.get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            slots = SIMDS * (gui / 8.0 / 4.0)
            k["kernel_quad_cycles"] = round(gui / 8.0 / 4.0, 1)
            if k["launch_ms"]:
                k["clock_ghz_under_profile"] = round(gui / 8.0 / (k["launch_ms"] * 1e-3) / 1e9, 3)
            if avg.get("SQ_INSTS_VALU@sq2"):
                k["valu_issue_frac"] = round(avg["SQ_INSTS_VALU@sq2"] / slots, 4)
            for src, dst in (("SQ_INSTS_SALU", "salu_issue_frac"), ("SQ_INSTS_SMEM", "smem_issue_frac"), ("SQ_INSTS_LDS", "lds_issue_frac"),
                             ("SQ_ACTIVE_INST_SCA", "scalar_busy_frac"), ("SQ_ACTIVE_INST_ANY", "any_busy_frac")):
                if src in avg:
                    k[dst] = round(avg[src] / slots, 4)
            if "SQ_INSTS_VMEM_RD" in avg:
                k["vmem_issue_frac"] = round((avg["SQ_INSTS_VMEM_RD"] + avg.get("SQ_INSTS_VMEM_WR", 0.0)) / slots, 4)
            # (pass sq has no clock of its own: SQ_ACTIVE_INST_VALU is scaled by the two passes' SQ_INSTS_VALU, equal up to run-to-run noise)
            if avg.get("SQ_ACTIVE_INST_VALU") and avg.get("SQ_INSTS_VALU") and avg.get("SQ_INSTS_VALU@sq2"):
                k["valu_busy_frac"] = round(avg["SQ_ACTIVE_INST_VALU"] * (avg["SQ_INSTS_VALU@sq2"] / avg["SQ_INSTS_VALU"]) / slots, 4)
        gui3, tot3 = avg.get("GRBM_GUI_ACTIVE@sq3", 0.0), avg.get("SQ_INSTS_VALU@sq3", 0.0)
        if gui3 and tot3:
            fast = avg.get("SQ_INSTS_VALU_ADD_F32", 0.0) + avg.get("SQ_INSTS_VALU_MUL_F32", 0.0) + avg.get("SQ_INSTS_VALU_FMA_F32", 0.0)
            trans = avg.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            other = max(tot3 - fast - trans, 0.0)
            slots3 = SIMDS * (gui3 / 8.0 / 4.0)
            k["valu_dual_rate_share"] = round(fast / tot3, 4)          # fp32 add / mul / fma among the vector instructions
            k["valu_mix"] = {"fma_f32": round(avg.get("SQ_INSTS_VALU_FMA_F32", 0.0) / tot3, 4), "mul_f32": round(avg.get("SQ_INSTS_VALU_MUL_F32", 0.0) / tot3, 4),
                             "add_f32": round(avg.get("SQ_INSTS_VALU_ADD_F32", 0.0) / tot3, 4), "trans_f32": round(trans / tot3, 4),
                             "cvt": round(avg.get("SQ_INSTS_VALU_CVT", 0.0) / tot3, 4), "int32": round(avg.get("SQ_INSTS_VALU_INT32", 0.0) / tot3, 4)}
            # issue slots the stream needs at the measured per-kind rates / slots the kernel had = the vector port's occupancy, as a RANGE: the
            # counters classify fp32 add / mul / fma (dual-rate), conversions (full-rate) and transcendentals (two slots); the rest is a mix of
            # full-rate forms (select, compare, min / max, lshl, bfe, mul_lo) and dual-rate ones (mov, and, xor, add_u32, lshr). Upper figure:
            # the rest at one slot each; lower: the rest at the dual rate. A port is full when the range brackets 1.
            cvt = min(avg.get("SQ_INSTS_VALU_CVT", 0.0), other)
            k["valu_slot_frac"] = round((fast / DUAL_RATE + other / 0.97 + 2.0 * trans) / slots3, 4)
            k["valu_slot_frac_lower"] = round(((fast + other - cvt) / DUAL_RATE + cvt / 0.97 + 2.0 * trans) / slots3, 4)
        if avg.get("TCP_TCC_READ_REQ_sum"):
            k["l2_read_requests_per_launch"] = avg["TCP_TCC_READ_REQ_sum"] * per_set  # 128-byte lines
            k["l1_accesses_per_launch"] = avg.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) * per_set
            if k["launch_ms"]:
                k["l2_read_requests_per_s"] = round(k["l2_read_requests_per_launch"] / (k["launch_ms"] * 1e-3), 0)
        kernels[fam] = k
    for first in ("shadeKernel", "missKernel"):  # bench.py times the last shading kernel and the accumulation as one interval
        if first in kernels and "accumulateKernel" in kernels:
            a, b = kernels[first], kernels["accumulateKernel"]
            # (bench.py gives the pair ONE stream-read figure: the add-back is taken against the pair's pooled FETCH_SIZE)
            fetch = a["fetch_bytes_raw"] + b["fetch_bytes_raw"]
            add_back = min(a["stream_read_bytes"], 2.0 * fetch) / 2.0
            kernels[first] = dict(a, stream_add_back_bytes=add_back, hbm_bytes_per_launch=fetch + add_back + a["write_bytes"] + b["write_bytes"],
                                  note="includes accumulateKernel's bytes")
    out = {
        "workload": args.workload, "frames_per_launch": args.frames, "source_sha": bench.kernel_source_sha(),
        "command": "rocprofv3 <pass flags> --output-format csv -d <dir> -- " + " ".join(["python3", "bench.py"] + cmd[2:]),
        "passes": {n: "rocprofv3 " + " ".join(f) for n, f in PASSES},
        "calibration": "profiles/r03_pmc_calibration.json",
        "definition": "hbm_bytes_per_launch = fabric-side bytes = FETCH_SIZE * 1024 (exact for this code's 64-byte gathers, 48-byte records and texel taps) "
                      "+ half of the kernel's coalesced 16-B-per-lane stream reads (the one pattern gfx950 tallies at half: 128-byte requests counted as 64) "
                      "+ WRITE_SIZE * 1024 (exact; scattered 16 + 4-byte stores really move a 32-byte sector each), averaged over the launches of the "
                      "non-counting kernel variants, each of which traces `frames_per_launch` frames. Infinity-Cache hits are included.",
        "kernels": kernels,
        "bench_line_under_stats": bench_lines.get("stats"),
    }
    path = os.path.join(args.out, f"pmc_{tag}.json")
    json.dump(out, open(path, "w"), indent=1)
    lines = [f"# {out['command']}", f"# source_sha {out['source_sha']}  frames per launch {args.frames}",
             f"{'kernel':24s} {'launches':>8s} {'launch_ms':>10s} {'FETCH MiB':>10s} {'+stream/2':>10s} {'WRITE MiB':>10s} {'fabric GB/s':>11s} {'frac':>6s} {'L2hit%':>7s} {'lanes':>6s} {'L2 rd req G/s':>13s}"]
    for fam, k in kernels.items():
        ms = k["launch_ms"] or 0.0
        gbs = k["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9 if ms else 0.0
        lines.append(f"{fam:24s} {k['launches_seen']:8d} {ms:10.3f} {k['fetch_bytes_raw'] / 2**20:10.1f} {k['stream_add_back_bytes'] / 2**20:10.1f} {k['write_bytes'] / 2**20:10.1f} {gbs:11.1f} "
                     f"{gbs / 8000.0:6.3f} {k['l2_hit_pct'] if k['l2_hit_pct'] is not None else 0:7.1f} {k.get('valu_lane_util', 0) or 0:6.3f} {(k.get('l2_read_requests_per_s') or 0) / 1e9:13.1f}")
    for f in glob.glob(os.path.join(args.out, f"{tag}_stats", "*", "*_kernel_stats.csv")):
        lines.append("")
        lines.append("# rocprofv3 --kernel-trace --stats (kernel_stats.csv)")
        lines.append(f"{'kernel':60s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
        for r in csv.DictReader(open(f)):
            lines.append(f"{r['Name'][:60]:60s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} {int(r['MinNs']) / 1e3:10.1f} {int(r['MaxNs']) / 1e3:10.1f} {float(r['Percentage']):7.2f}")
    open(os.path.join(args.out, f"pmc_{tag}.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
