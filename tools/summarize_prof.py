"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into the small summaries kept under profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof profiles/r01_sponza1080
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE under-reports wide (16 B/lane)
reads by 2x (MI355X_MICROARCH.md, HBM section), so `fetch_corrected` doubles it — an upper bound for narrower loads.
"""
import collections, csv, glob, json, os, re, sys


def short(name):
    m = re.search(r"(tracePacketKernel<\w+, \w+>|tracePacketKernel<\w+>|traceKernel<\w+>|shadeKernel<\w+, \w+>|generateKernel|accumulateKernel|renderReferenceOrder|scaleTextureKernel|__amd_rocclr_\w+)", name)
    return m.group(1) if m else name[:40]


def main(src, dst):
    out = {}
    lines = []
    for f in glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Name"])
            out.setdefault("kernel_stats", {})[k] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "min_ns": int(r["MinNs"]),
                                                    "max_ns": int(r["MaxNs"]), "pct": float(r["Percentage"])}
    lines.append("## rocprofv3 --kernel-trace --stats   (python3 bench.py --no-cpu-baseline --no-single-call --steps 128 --warmup 64: launch sets of 64 frames ,")
    lines.append("## so per-frame time of a kernel = max_us / 64; the 1-frame launches of the counter pre-pass and the warm-up set pull the averages down)")
    lines.append(f"{'kernel':42s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
    for k, v in sorted(out.get("kernel_stats", {}).items(), key=lambda kv: -kv[1]["pct"]):
        lines.append(f"{k:42s} {v['calls']:6d} {v['avg_ns']/1e3:10.1f} {v['min_ns']/1e3:10.1f} {v['max_ns']/1e3:10.1f} {v['pct']:7.2f}")
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(src, d, "*", "*_counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines.append("")
    lines.append("## rocprofv3 --pmc (separate passes: FETCH_SIZE | WRITE_SIZE TCC_HIT_sum TCC_MISS_sum), per launch averages")
    lines.append(f"{'kernel':42s} {'n':>4s} {'FETCH_MiB':>10s} {'x2 corr':>10s} {'WRITE_MiB':>10s} {'L2 hit%':>8s}")
    out["pmc"] = {}
    for k, cs in sorted(pmc.items()):
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        n = max(len(v) for v in cs.values())
        fetch = avg.get("FETCH_SIZE", 0.0) * 1024
        write = avg.get("WRITE_SIZE", 0.0) * 1024
        hit, miss = avg.get("TCC_HIT_sum", 0.0), avg.get("TCC_MISS_sum", 0.0)
        hr = 100.0 * hit / (hit + miss) if hit + miss else 0.0
        out["pmc"][k] = {"launches": n, "fetch_bytes": fetch, "fetch_corrected_bytes": 2 * fetch, "write_bytes": write, "l2_hit_pct": hr}
        lines.append(f"{k:42s} {n:4d} {fetch/2**20:10.2f} {2*fetch/2**20:10.2f} {write/2**20:10.2f} {hr:8.1f}")
    for name in ("bench_trace.log", "bench_pmc_fetch.log"):
        p = os.path.join(src, name)
        if os.path.exists(p):
            js = [l for l in open(p) if l.startswith("{")]
            if js:
                lines.append("")
                lines.append(f"## bench.py JSON line under the profiler ({name})")
                lines.append(js[-1].strip())
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    open(dst + "_summary.txt", "w").write("\n".join(lines) + "\n")
    json.dump(out, open(dst + "_summary.json", "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
