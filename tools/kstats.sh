#!/bin/bash
# per-kernel average durations of the bench workload under given env knobs: tools/kstats.sh "ENV=.. ENV=.." tag
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo "$cfg" | tr ' =' '__'); out=gpurun_out/kstats/$tag; mkdir -p $out
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 16 --warmup 2 > $out/log.txt 2>&1
  echo "== $cfg"; python3 - <<PY
import csv, glob, re
for f in glob.glob("$out/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(traceKernel<\w+>|shadeKernel<\w+, \w+>|generateKernel)", r["Name"])
        if m and not m.group(1).endswith("true>"): print("   %-28s calls %4s avg_us %9.1f" % (m.group(1), r["Calls"], float(r["AverageNs"])/1e3))
PY
done
