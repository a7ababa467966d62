#!/bin/bash
# one rank's share of an N-way tile split, emulated on one GPU: wall ms/frame and per-kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
mkdir -p gpurun_out/tile
for cfg in "OKRT_BENCH_TILE=8,0,8 OKRT_BENCH_FPL=64" "OKRT_BENCH_FPL=8" "OKRT_BENCH_TILE=8,0,2 OKRT_BENCH_FPL=64" "OKRT_BENCH_TILE=8,0,4 OKRT_BENCH_FPL=64"; do
  tag=$(echo "$cfg" | tr ' =,' '___'); out=gpurun_out/tile/$tag; rm -rf $out; mkdir -p $out
  echo "== $cfg"
  env $cfg python3 bench.py --no-cpu-baseline --steps 128 --warmup 64 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('   wall ms/step', d['ms_per_step'], 'event', d['config']['gpu_event_ms_per_step'])"
  env $cfg rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu-baseline --steps 128 --warmup 64 > $out/log.txt 2>&1
  python3 - <<PY
import csv, glob, re
tot = 0
for f in glob.glob("$out/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(tracePacketKernel<\w+>|traceKernel<\w+>|shadeKernel<\w+, \w+>|generateKernel|accumulateKernel)", r["Name"])
        if m and not m.group(1).endswith("true>"):
            print("   %-28s calls %4s avg_us %9.1f total_ms %8.2f" % (m.group(1), r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6)); tot += float(r["TotalDurationNs"])/1e6
print("   kernel total ms", round(tot, 2))
PY
done
