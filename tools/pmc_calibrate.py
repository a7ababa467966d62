#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE against KNOWN bytes, pattern by pattern (tools/calib/pmc_calibrate.hip), on the GPU box:

    python3 tools/pmc_calibrate.py --out gpurun_out/pmc_cal        (via gpurun; then copy the .json/.txt under profiles/)

Three rocprofv3 passes of the same binary (counters never share a run with --stats; FETCH_SIZE and WRITE_SIZE do not fit
one pass): --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE. For every kernel x table size the LAST launch
(caches warm: the 64 MiB table is Infinity-Cache resident by then) gives
    bytes_per_counted_byte = algorithmic bytes / (counter x 1024)
which tools/pmc_traffic.py then uses per kernel family instead of a blanket x2, and the rate the pattern reaches."""
import argparse
import collections
import csv
import glob
import json
import os
import subprocess

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(REPO, "tools", "calib", "pmc_calibrate")
SRC = BIN + ".hip"
PASSES = [("stats", ["--kernel-trace", "--stats"]), ("fetch", ["--pmc", "FETCH_SIZE", "--kernel-trace"]), ("write", ["--pmc", "WRITE_SIZE", "--kernel-trace"])]
CASES = ["stream16_read", "gather64_read", "gather48_read", "texel_taps_read", "stream16_write", "scatter20_write", "scatter16_write"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/pmc_cal")
    ap.add_argument("--launches", type=int, default=3)
    args = ap.parse_args()
    out = os.path.abspath(args.out)
    os.makedirs(out, exist_ok=True)
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < os.path.getmtime(SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", SRC, "-o", BIN], check=True)
    env = dict(os.environ, TMPDIR="/tmp")
    plain = subprocess.run([BIN, str(args.launches)], capture_output=True, text=True, check=True).stdout  # un-profiled times
    rates = {}
    for r in csv.DictReader(plain.splitlines()):
        rates[(r["case"], int(r["table_bytes"]))] = (int(r["algorithmic_bytes"]), float(r["ms"]), float(r["GBps"]))  # last launch wins
    for name, flags in PASSES:
        d = os.path.join(out, name)
        with open(os.path.join(out, name + ".log"), "w") as lf:
            subprocess.run(["rocprofv3"] + flags + ["--output-format", "csv", "-d", d, "--", BIN, str(args.launches)], stdout=lf, stderr=subprocess.STDOUT, env=env,
                           cwd="/tmp", check=True)
    # per kernel name, dispatches in order: launches x (64 MiB), then launches x (1 GiB)
    ctr = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("fetch", "write"):
        for f in glob.glob(os.path.join(out, name, "*", "*_counter_collection.csv")):
            rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
            for r in rows:
                k = r["Kernel_Name"].split("(")[0]
                if k in CASES:
                    ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    result = {"definition": "multiplier = algorithmic bytes of the launch (every byte of the table touched exactly once) / (counter x 1024); last of "
                            f"{args.launches} launches per case (caches warm)", "cases": {}}
    lines = [f"{'pattern':18s} {'table':>8s} {'alg MiB':>9s} {'FETCH MiB':>10s} {'WRITE MiB':>10s} {'alg/FETCH':>9s} {'alg/WRITE':>9s} {'GB/s':>8s}"]
    for case in CASES:
        for ti, table in enumerate((64 << 20, 1 << 30)):
            alg, ms, gbps = rates[(case, table)]
            idx = ti * args.launches + args.launches - 1
            fetch = ctr[case]["FETCH_SIZE"][idx] * 1024.0 if len(ctr[case]["FETCH_SIZE"]) > idx else None
            write = ctr[case]["WRITE_SIZE"][idx] * 1024.0 if len(ctr[case]["WRITE_SIZE"]) > idx else None
            is_read = case.endswith("_read")
            mult = (alg / fetch if fetch else None) if is_read else (alg / write if write else None)
            result["cases"][f"{case}@{table >> 20}MiB"] = {"algorithmic_bytes": alg, "fetch_bytes_counted": fetch, "write_bytes_counted": write,
                                                           "multiplier": round(mult, 4) if mult else None, "ms": ms, "GBps": gbps}
            lines.append(f"{case:18s} {table >> 20:6d}Mi {alg / 2**20:9.1f} {(fetch or 0) / 2**20:10.1f} {(write or 0) / 2**20:10.1f} "
                         f"{(alg / fetch) if fetch else 0:9.3f} {(alg / write) if write else 0:9.3f} {gbps:8.1f}")
    for f in glob.glob(os.path.join(out, "stats", "*", "*_kernel_stats.csv")):
        lines += ["", "# rocprofv3 --kernel-trace --stats (all launches of both table sizes averaged)", f"{'kernel':24s} {'calls':>6s} {'avg_us':>10s}"]
        for r in csv.DictReader(open(f)):
            lines.append(f"{r['Name'].split('(')[0][:24]:24s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f}")
    json.dump(result, open(os.path.join(out, "pmc_calibration.json"), "w"), indent=1)
    open(os.path.join(out, "pmc_calibration.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
