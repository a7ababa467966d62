#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun). Kernel-trace and PMC passes are separate runs.
set -e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-single-call --steps 128 --warmup 64 > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --no-cpu-baseline --no-single-call --steps 64 --warmup 64 > $OUT/bench_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 bench.py --no-cpu-baseline --no-single-call --steps 64 --warmup 64 > $OUT/bench_pmc_write.log 2>&1
find $OUT -name "*.csv" | head -50
