#!/bin/bash
# kernel trace of the device builder (tools/build_bench.py): tools/profile_build.sh [scene] [W H]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
out=gpurun_out/prof_build; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/build_bench.py "$@" > $out/log.txt 2>&1
cat $out/log.txt | grep -v amdgpu.ids
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/*/*_kernel_stats.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
keep = ("bakeKernel", "mortonKernel", "nearestKernel", "mergeKernel", "collapseKernel", "radix", "onesweep", "partition", "select", "lookback", "scan", "histogram")
print("%-70s %6s %10s %10s" % ("kernel", "calls", "avg_us", "total_ms"))
for r in rows:
    if any(k in r["Name"] for k in keep):
        print("%-70s %6s %10.1f %10.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
